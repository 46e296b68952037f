"""lamehip -- thin ctypes host binding over liblamehip.so (the C ABI in
include/lamehip.h).  It mirrors the reference's lame.h call sequence
(lame_init -> lame_set_* -> lame_init_params -> lame_encode_buffer ->
lame_encode_flush -> lame_close) and adds the batch entry points.  There is no
Python or CPU implementation of the encode path here: if the HIP library is
missing, importing the binding's loader fails loudly.
"""
import ctypes as C
import os

import numpy as np

from .types import LhConfig, LhFrameOut, LhTables  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LAMEHIP_LIB", os.path.join(_HERE, "liblamehip.so"))   # LAMEHIP_LIB: profiling build

ERR_NODEVICE = -10
STEREO, JOINT_STEREO = 0, 1

_lib = None


def load_library():
    """Load liblamehip.so (built in-tree by __graft_entry__.build()); raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "liblamehip.so not found at %s: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the encode path has no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.lame_init.restype = C.c_void_p
    for name in ("lame_set_in_samplerate", "lame_set_num_channels", "lame_set_brate", "lame_set_mode",
                 "lame_set_quality", "lame_set_VBR", "lame_set_bWriteVbrTag", "lame_set_out_samplerate"):
        getattr(lib, name).argtypes = [C.c_void_p, C.c_int]
    lib.lame_init_params.argtypes = [C.c_void_p]
    lib.lame_encode_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    lib.lame_encode_flush.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.lame_close.argtypes = [C.c_void_p]
    lib.lame_get_lametag_frame.restype = C.c_size_t
    lib.lame_get_lametag_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.lame_get_frameNum.argtypes = [C.c_void_p]
    lib.lamehip_last_error.restype = C.c_char_p
    lib.lamehip_get_config.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.lamehip_get_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.lamehip_batch_create.restype = C.c_void_p
    lib.lamehip_batch_create.argtypes = [C.c_void_p, C.c_int, C.c_long]
    lib.lamehip_batch_create_on.restype = C.c_void_p
    lib.lamehip_batch_create_on.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_long]
    lib.lamehip_set_device.argtypes = [C.c_void_p, C.c_int]
    lib.lamehip_batch_destroy.argtypes = [C.c_void_p]
    lib.lamehip_batch_set_pcm.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_long]
    lib.lamehip_batch_set_pcm_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_long]
    lib.lamehip_batch_set_length.argtypes = [C.c_void_p, C.c_int, C.c_long]
    lib.lamehip_batch_pcm_device_ptr.restype = C.c_void_p
    lib.lamehip_batch_pcm_device_ptr.argtypes = [C.c_void_p]
    lib.lamehip_batch_encode.argtypes = [C.c_void_p]
    lib.lamehip_batch_pack_all.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_void_p]
    lib.lamehip_batch_sync.argtypes = [C.c_void_p]
    lib.lamehip_batch_reset.argtypes = [C.c_void_p]
    lib.lamehip_batch_frames.argtypes = [C.c_void_p, C.c_int]
    lib.lamehip_batch_pack.restype = C.c_long
    lib.lamehip_batch_pack.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_long]
    lib.lamehip_batch_get_frames.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    lib.lamehip_batch_get_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    lib.lamehip_batch_last_kernel_ms.restype = C.c_float
    lib.lamehip_batch_last_kernel_ms.argtypes = [C.c_void_p]
    lib.lamehip_batch_last_kernel_parts_ms.restype = C.c_int
    lib.lamehip_batch_last_kernel_parts_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.lamehip_batch_kernel_waves.argtypes = [C.c_void_p]
    lib.lamehip_batch_last_windows.restype = C.c_int
    lib.lamehip_batch_last_windows.argtypes = [C.c_void_p]
    _lib = lib
    return lib


def last_error():
    return load_library().lamehip_last_error().decode()


class Encoder:
    """One stream behind the lame.h call sequence."""

    def __init__(self, samplerate=44100, brate=128, mode=None, quality=None, require_device=True, write_tag=False,
                 vbr_q=None, out_samplerate=0, abr=None, channels=2, device=None, vbr_mode=4, error_protection=False):
        self.lib = load_library()
        self.h = C.c_void_p(self.lib.lame_init())
        if device is not None:      # HIP device of the handle's own launches (lamehip_set_device)
            self.lib.lamehip_set_device(self.h, int(device))
        self.lib.lame_set_in_samplerate(self.h, samplerate)
        if out_samplerate:
            self.lib.lame_set_out_samplerate(self.h, out_samplerate)
        self.lib.lame_set_num_channels(self.h, channels)  # 1: mono (only the left buffer is read)
        self.channels = channels
        if abr is not None:         # ABR at a mean bitrate of `abr' kb/s (the reference's --abr n)
            self.lib.lame_set_VBR(self.h, 3)
            self.lib.lame_set_VBR_mean_bitrate_kbps(self.h, abr)
        elif vbr_q is None:
            self.lib.lame_set_brate(self.h, brate)
        else:                       # VBR at quality vbr_q (the reference's -V n): 4 vbr_mtrh, 1 vbr_mt, 2 vbr_rh (--vbr-old)
            self.lib.lame_set_VBR(self.h, vbr_mode)
            self.lib.lame_set_VBR_q(self.h, vbr_q)
        self.lib.lame_set_bWriteVbrTag(self.h, 1 if write_tag else 0)
        if error_protection:        # CRC-16 behind the header (the reference's -p)
            self.lib.lame_set_error_protection(self.h, 1)
        if mode is not None:
            self.lib.lame_set_mode(self.h, mode)
        if quality is not None:
            self.lib.lame_set_quality(self.h, quality)
        self.rc = self.lib.lame_init_params(self.h)
        if self.rc == ERR_NODEVICE and not require_device:
            return
        if self.rc != 0:
            msg = last_error()
            self.close()
            raise RuntimeError("lame_init_params failed (%d): %s" % (self.rc, msg))

    def config(self):
        c = LhConfig()
        assert self.lib.lamehip_get_config(self.h, C.byref(c), C.sizeof(c)) == C.sizeof(c)
        return c

    def tables(self):
        t = LhTables()
        assert self.lib.lamehip_get_tables(self.h, C.byref(t), C.sizeof(t)) == C.sizeof(t)
        return t

    def encode(self, left, right=None):
        left = np.ascontiguousarray(left, dtype=np.int16)
        right = left if right is None else np.ascontiguousarray(right, dtype=np.int16)   # mono: not read
        n = len(left)
        buf = C.create_string_buffer(int(1.25 * n) + 7200)
        k = self.lib.lame_encode_buffer(self.h, left.ctypes.data, right.ctypes.data, n, buf, len(buf))
        if k < 0:
            raise RuntimeError("lame_encode_buffer failed (%d): %s" % (k, last_error()))
        return buf.raw[:k]

    def flush(self):
        buf = C.create_string_buffer(4 * 7200)
        k = self.lib.lame_encode_flush(self.h, buf, len(buf))
        if k < 0:
            raise RuntimeError("lame_encode_flush failed (%d): %s" % (k, last_error()))
        return buf.raw[:k]

    def lametag_frame(self):
        """Final Xing/Info + LAME tag frame (b"" when the tag is off)."""
        buf = C.create_string_buffer(2880)
        k = self.lib.lame_get_lametag_frame(self.h, buf, len(buf))
        return buf.raw[:k]

    def close(self):
        if self.h:
            self.lib.lame_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """B independent streams with common settings, encoded by one kernel launch."""

    def __init__(self, enc, nstreams, capacity, device=None):
        """device: HIP device index the batch lives on (lamehip_batch_create_on); None = the calling
        thread's current device."""
        self.lib = enc.lib
        self.enc = enc
        self.n = nstreams
        self.capacity = capacity
        if device is None:
            self.b = C.c_void_p(self.lib.lamehip_batch_create(enc.h, nstreams, capacity))
        else:
            self.b = C.c_void_p(self.lib.lamehip_batch_create_on(int(device), enc.h, nstreams, capacity))
        if not self.b:
            raise RuntimeError("lamehip_batch_create failed: %s" % last_error())

    def set_pcm(self, s, left, right=None):
        left = np.ascontiguousarray(left, dtype=np.int16)
        right = left if right is None else np.ascontiguousarray(right, dtype=np.int16)   # mono: not read
        rc = self.lib.lamehip_batch_set_pcm(self.b, s, left.ctypes.data, right.ctypes.data, len(left))
        if rc:
            raise RuntimeError("lamehip_batch_set_pcm failed (%d): %s" % (rc, last_error()))

    def set_pcm_device(self, s, dev_left_ptr, dev_right_ptr, n):
        rc = self.lib.lamehip_batch_set_pcm_device(self.b, s, dev_left_ptr, dev_right_ptr, n)
        if rc:
            raise RuntimeError("lamehip_batch_set_pcm_device failed (%d): %s" % (rc, last_error()))

    # ---- incremental use: lame_encode_buffer semantics for all streams of the batch ----
    def append(self, s, left, right=None):
        """Stage more samples of stream s (any number, also 0); encode_available() moves everything
        staged to the GPU with one copy."""
        left = np.ascontiguousarray(left, dtype=np.int16)
        right = left if right is None else np.ascontiguousarray(right, dtype=np.int16)
        self.lib.lamehip_batch_append.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        rc = self.lib.lamehip_batch_append(self.b, s, left.ctypes.data, right.ctypes.data, len(left))
        if rc:
            raise RuntimeError("lamehip_batch_append failed (%d): %s" % (rc, last_error()))

    def encode_available(self):
        """Encode every stream's newly complete frames (one launch); returns how many there were."""
        self.lib.lamehip_batch_encode_available.argtypes = [C.c_void_p]
        n = self.lib.lamehip_batch_encode_available(self.b)
        if n < 0:
            raise RuntimeError("lamehip_batch_encode_available failed (%d): %s" % (n, last_error()))
        return n

    def finish(self):
        """lame_encode_flush for every stream."""
        self.lib.lamehip_batch_finish.argtypes = [C.c_void_p]
        n = self.lib.lamehip_batch_finish(self.b)
        if n < 0:
            raise RuntimeError("lamehip_batch_finish failed (%d): %s" % (n, last_error()))
        return n

    def drain(self, s, cap=1 << 20):
        """Bytes stream s has produced since its last drain."""
        self.lib.lamehip_batch_drain.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        buf = C.create_string_buffer(cap)
        k = self.lib.lamehip_batch_drain(self.b, s, buf, cap)
        if k < 0:
            raise RuntimeError("lamehip_batch_drain: buffer of %d bytes too small" % cap)
        return buf.raw[:k]

    def set_length(self, s, n):
        assert self.lib.lamehip_batch_set_length(self.b, s, n) == 0

    def pcm_device_ptr(self):
        return self.lib.lamehip_batch_pcm_device_ptr(self.b)

    def set_device_packing(self, on=True):
        """Let the kernel assemble the MP3 bytes in HBM (get_bytes) instead of leaving it to the host packer."""
        self.lib.lamehip_batch_set_device_packing.argtypes = [C.c_void_p, C.c_int]
        assert self.lib.lamehip_batch_set_device_packing(self.b, 1 if on else 0) == 0

    def get_bytes(self, s):
        self.lib.lamehip_batch_get_bytes.restype = C.c_long
        self.lib.lamehip_batch_get_bytes.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_long]
        cap = (self.frames(s) + 2) * 1500
        buf = C.create_string_buffer(cap)
        k = self.lib.lamehip_batch_get_bytes(self.b, s, buf, cap)
        if k < 0:
            raise RuntimeError("lamehip_batch_get_bytes failed (%d): %s" % (k, last_error()))
        return buf.raw[:k]

    def get_bytes_tagged(self, s):
        self.lib.lamehip_batch_get_bytes_tagged.restype = C.c_long
        self.lib.lamehip_batch_get_bytes_tagged.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_long]
        cap = (self.frames(s) + 4) * 1500
        buf = C.create_string_buffer(cap)
        k = self.lib.lamehip_batch_get_bytes_tagged(self.b, s, buf, cap)
        if k < 0:
            raise RuntimeError("lamehip_batch_get_bytes_tagged failed (%d): %s" % (k, last_error()))
        return buf.raw[:k]

    def get_bytes_all(self, stride=None):
        """(buffer [B, stride] uint8, sizes) of the device-packed streams."""
        if stride is None:
            stride = (max(self.frames(s) for s in range(self.n)) + 2) * 1500
        out = np.empty((self.n, stride), dtype=np.uint8)
        sizes = np.zeros(self.n, dtype=np.int64)
        self.lib.lamehip_batch_get_bytes_all.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
        rc = self.lib.lamehip_batch_get_bytes_all(self.b, out.ctypes.data, stride, sizes.ctypes.data)
        if rc:
            raise RuntimeError("lamehip_batch_get_bytes_all failed (%d): %s" % (rc, last_error()))
        return out, sizes

    def pcm_host(self):
        """The pinned host mirror of the s16 pool as an int16 array [streams, 2, capacity] (no copy): fill it, then
        set_length(s, n) + mark_pcm(s); upload() / encode() move it to HBM with one asynchronous copy."""
        self.lib.lamehip_batch_pcm_host_ptr.restype = C.c_void_p
        self.lib.lamehip_batch_pcm_host_ptr.argtypes = [C.c_void_p]
        p = self.lib.lamehip_batch_pcm_host_ptr(self.b)
        if not p:
            raise RuntimeError("no pinned mirror for this batch: %s" % last_error())
        buf = (C.c_int16 * (self.n * 2 * self.capacity)).from_address(p)
        return np.frombuffer(buf, dtype=np.int16).reshape(self.n, 2, self.capacity)

    def mark_pcm(self, s):
        assert self.lib.lamehip_batch_mark_pcm(self.b, s) == 0

    def upload(self):
        rc = self.lib.lamehip_batch_upload(self.b)
        if rc:
            raise RuntimeError("lamehip_batch_upload failed (%d): %s" % (rc, last_error()))

    def fetch(self):
        """Start the device-packed bytes' way back to pinned host memory (asynchronous, behind the kernel)."""
        rc = self.lib.lamehip_batch_fetch(self.b)
        if rc:
            raise RuntimeError("lamehip_batch_fetch failed (%d): %s" % (rc, last_error()))

    def bytes_view(self, s):
        """Stream s's finished bytes in the batch's pinned buffer (waits for fetch()); a uint8 view, no copy."""
        self.lib.lamehip_batch_bytes_ptr.restype = C.c_long
        self.lib.lamehip_batch_bytes_ptr.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        p = C.c_void_p()
        k = self.lib.lamehip_batch_bytes_ptr(self.b, s, C.byref(p))
        if k < 0:
            raise RuntimeError("lamehip_batch_bytes_ptr failed (%d): %s" % (k, last_error()))
        if k == 0:
            return np.zeros(0, dtype=np.uint8)
        return np.frombuffer((C.c_uint8 * k).from_address(p.value), dtype=np.uint8)

    def encode(self, sync=True):
        rc = self.lib.lamehip_batch_encode(self.b)
        if rc:
            raise RuntimeError("lamehip_batch_encode failed (%d): %s" % (rc, last_error()))
        if sync:
            self.sync()

    def reserve(self):
        """Allocate the launch's buffers now (payload, analysis pool) instead of in the first encode()."""
        self.lib.lamehip_batch_reserve.argtypes = [C.c_void_p]
        rc = self.lib.lamehip_batch_reserve(self.b)
        if rc:
            raise RuntimeError("lamehip_batch_reserve failed (%d): %s" % (rc, last_error()))

    def sync(self):
        rc = self.lib.lamehip_batch_sync(self.b)
        if rc:
            raise RuntimeError("lamehip_batch_sync failed (%d): %s" % (rc, last_error()))

    def reset(self):
        assert self.lib.lamehip_batch_reset(self.b) == 0

    def kernel_ms(self):
        return float(self.lib.lamehip_batch_last_kernel_ms(self.b))

    def kernel_parts_ms(self):
        """The last launch kernel by kernel: (split, [analysis, sub-band, encode] in ms); split = False: one fused kernel."""
        parts = (C.c_float * 3)()
        split = self.lib.lamehip_batch_last_kernel_parts_ms(self.b, parts)
        return bool(split == 1), [float(parts[0]), float(parts[1]), float(parts[2])]

    def windows(self):
        """Sub-launches of the last launch: 1, or the frame windows the split pipeline worked through (lamehip.h)."""
        return int(self.lib.lamehip_batch_last_windows(self.b))

    def kernel_waves(self):
        return int(self.lib.lamehip_batch_kernel_waves(self.b))

    def frames(self, s):
        return self.lib.lamehip_batch_frames(self.b, s)

    def get_frames(self, s):
        n = self.frames(s)
        arr = (LhFrameOut * n)()
        got = self.lib.lamehip_batch_get_frames(self.b, s, arr, n)
        if got != n:
            raise RuntimeError("lamehip_batch_get_frames failed (%d): %s" % (got, last_error()))
        return arr

    def pack(self, s):
        n = self.frames(s)
        buf = C.create_string_buffer(n * 1500 + 8192)
        k = self.lib.lamehip_batch_pack(self.b, s, buf, len(buf))
        if k < 0:
            raise RuntimeError("lamehip_batch_pack failed (%d): %s" % (k, last_error()))
        return buf.raw[:k]

    def pack_tagged(self, s):
        """Complete file image of stream s: final tag frame + audio frames."""
        n = self.frames(s)
        buf = C.create_string_buffer(n * 1500 + 8192 + 2880)
        self.lib.lamehip_batch_pack_tagged.restype = C.c_long
        self.lib.lamehip_batch_pack_tagged.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_long]
        k = self.lib.lamehip_batch_pack_tagged(self.b, s, buf, len(buf))
        if k < 0:
            raise RuntimeError("lamehip_batch_pack_tagged failed (%d): %s" % (k, last_error()))
        return buf.raw[:k]

    def pack_all(self, nthreads=0, as_bytes=True):
        """Pack every stream with `nthreads' host threads (0 = min(32, cores)).  Returns the list of
        mp3 byte strings, or (buffer, stride, sizes) when as_bytes is False."""
        import os
        nthreads = nthreads or min(32, os.cpu_count() or 1)
        stride = max(self.frames(s) for s in range(self.n)) * 1500 + 8192
        buf = np.empty(self.n * stride, dtype=np.uint8)
        sizes = np.zeros(self.n, dtype=np.int64)
        rc = self.lib.lamehip_batch_pack_all(self.b, nthreads, buf.ctypes.data, stride, sizes.ctypes.data)
        if rc != 0 or (sizes < 0).any():
            raise RuntimeError("lamehip_batch_pack_all failed (%d): %s" % (rc, last_error()))
        if not as_bytes:
            return buf, stride, sizes
        return [buf[s * stride: s * stride + int(sizes[s])].tobytes() for s in range(self.n)]

    def close(self):
        if self.b:
            self.lib.lamehip_batch_destroy(self.b)
            self.b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_streams(total_streams, world_size, rank):
    """Static sharding of a batch of independent streams over the GPUs of a node:
    rank r of world_size owns the contiguous block [lo, hi) of global stream indices; blocks differ
    by at most one stream and cover the batch exactly.  No collective is involved anywhere on the
    data path -- a stream's state never leaves its device."""
    if world_size <= 0 or not (0 <= rank < world_size) or total_streams < 0:
        raise ValueError("shard_streams(%r, %r, %r)" % (total_streams, world_size, rank))
    base, extra = divmod(total_streams, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)
