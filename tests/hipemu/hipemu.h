/*
 * hipemu.h -- TEST TOOL: a tiny CPU emulator for the subset of HIP that
 * csrc/lh_kernels.hip uses, so that the kernel SOURCE can be executed and
 * debugged in the build container (which has no GPU) before GPU minutes are
 * spent.  Every GPU thread becomes a ucontext fiber; __syncthreads() and the
 * wave-level primitives of lh_wave.h become fiber barriers.  Blocks run one
 * after another on the calling OS thread.
 *
 * This is NOT a product path: nothing under deprecated-lame-mirror_amd/ includes
 * it except through the LH_EMU switch that only tests/hipemu/Makefile sets, and
 * the product library is always built by hipcc for gfx950.
 */
#ifndef HIPEMU_H
#define HIPEMU_H

#include <ucontext.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

struct hipemu_dim3 {
    unsigned x, y, z;
};

struct hipemu_fiber {
    ucontext_t ctx;
    char   *stack;
    hipemu_dim3 tid;
    int     done;
};

struct hipemu_barrier {
    int     count;
    unsigned generation;
};

struct hipemu_state {
    std::vector < hipemu_fiber > fibers;
    ucontext_t sched;
    int     cur;
    hipemu_dim3 bid, bdim, gdim;
    hipemu_barrier block_bar;
    hipemu_barrier wave_bar[64];       /* up to 64 waves per block */
    uint64_t xchg[64][64];             /* per wave exchange area for shuffles / reductions */
    std::function < void () > body;
    unsigned long long nswitch;
};

extern hipemu_state *hipemu_g;

#define threadIdx (hipemu_g->fibers[hipemu_g->cur].tid)
#define blockIdx  (hipemu_g->bid)
#define blockDim  (hipemu_g->bdim)
#define gridDim   (hipemu_g->gdim)

static inline void
hipemu_yield(void)
{
    hipemu_state *g = hipemu_g;
    g->nswitch++;
    swapcontext(&g->fibers[g->cur].ctx, &g->sched);
}

static inline void
hipemu_barrier_wait(hipemu_barrier * b, int n)
{
    unsigned gen = b->generation;
    if (++b->count == n) {
        b->count = 0;
        b->generation++;
    }
    else {
        while (b->generation == gen)
            hipemu_yield();
    }
}

static inline void
__syncthreads(void)
{
    hipemu_barrier_wait(&hipemu_g->block_bar, (int) hipemu_g->bdim.x);
}

static inline int
hipemu_lane(void)
{
    return (int) (threadIdx.x & 63);
}

static inline int
hipemu_wave(void)
{
    return (int) (threadIdx.x >> 6);
}

static inline void
hipemu_wave_sync(void)
{
    hipemu_barrier_wait(&hipemu_g->wave_bar[hipemu_wave()], 64);
}

/* all-lanes exchange: every lane deposits v, then may read any lane's value */
static inline const uint64_t *
hipemu_wave_exchange(uint64_t v)
{
    uint64_t *x = hipemu_g->xchg[hipemu_wave()];
    hipemu_wave_sync();         /* previous readers are done */
    x[hipemu_lane()] = v;
    hipemu_wave_sync();
    return x;
}

void    hipemu_run(hipemu_dim3 grid, hipemu_dim3 block, std::function < void () > body);

#endif
