/* hipemu.cpp -- TEST TOOL: fiber scheduler of the HIP-subset CPU emulator (see hipemu.h). */
#include "hipemu.h"

hipemu_state *hipemu_g = 0;

static void
fiber_entry(void)
{
    hipemu_state *g = hipemu_g;
    g->body();
    g->fibers[g->cur].done = 1;
    swapcontext(&g->fibers[g->cur].ctx, &g->sched);
}

void
hipemu_run(hipemu_dim3 grid, hipemu_dim3 block, std::function < void () > body)
{
    static hipemu_state st;
    const size_t STACK = 512 * 1024;
    hipemu_g = &st;
    st.body = body;
    st.gdim = grid;
    st.bdim = block;
    if (st.fibers.size() != block.x) {
        for (auto & f:st.fibers)
            free(f.stack);
        st.fibers.assign(block.x, hipemu_fiber());
        for (auto & f:st.fibers)
            f.stack = (char *) malloc(STACK);
    }
    /* (grids of one or two dimensions) */
    unsigned const gy = grid.y ? grid.y : 1;
    for (unsigned bb = 0; bb < grid.x * gy; bb++) {
        unsigned const b = bb % grid.x;
        st.bid.x = b;
        st.bid.y = bb / grid.x;
        st.bid.z = 0;
        memset(&st.block_bar, 0, sizeof(st.block_bar));
        memset(st.wave_bar, 0, sizeof(st.wave_bar));
        for (unsigned t = 0; t < block.x; t++) {
            hipemu_fiber & f = st.fibers[t];
            f.tid.x = t;
            f.tid.y = f.tid.z = 0;
            f.done = 0;
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack;
            f.ctx.uc_stack.ss_size = STACK;
            f.ctx.uc_link = 0;
            makecontext(&f.ctx, fiber_entry, 0);
        }
        unsigned remaining = block.x;
        while (remaining) {
            for (unsigned t = 0; t < block.x; t++) {
                if (st.fibers[t].done)
                    continue;
                st.cur = (int) t;
                swapcontext(&st.sched, &st.fibers[t].ctx);
                if (st.fibers[t].done)
                    remaining--;
            }
        }
    }
}
