import sys, time, os
sys.path.insert(0,'/root/repo'); os.chdir('/root/repo')
import __graft_entry__ as g
t=time.time(); g.smoke(); print('smoke time', time.time()-t)
