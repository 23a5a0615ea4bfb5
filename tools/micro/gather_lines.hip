// Micro-benchmark: how many random cache lines per second can the chip fetch?
// Every lane issues MLP independent 16-byte loads per iteration, each from a random 64-byte-aligned
// (or 128-byte-aligned) slot of a table.  This is the request-rate ceiling for the triangle and
// deep-node fetches of the trace kernel (one lane = one line).
// build: hipcc -O3 --offload-arch=gfx950 -o build/gather_lines tools/micro/gather_lines.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// BOTH: every slot is a 128-byte line and the lane reads 16 bytes from each 64-byte half of it
template <int MLP, bool BOTH = false>
__global__ void __launch_bounds__(256) gather(const uint4* __restrict__ table, unsigned n_slots, unsigned stride16,
                                              int iters, unsigned* __restrict__ out) {
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    unsigned acc = 0, h = hash32(gid + 1u);
    for (int it = 0; it < iters; ++it) {
        uint4 v[MLP], w[MLP];
#pragma unroll
        for (int k = 0; k < MLP; ++k) {
            h = hash32(h + 0x9E3779B9u);
            v[k] = table[(size_t)(h % n_slots) * stride16];
            if (BOTH) w[k] = table[(size_t)(h % n_slots) * stride16 + 4];
        }
#pragma unroll
        for (int k = 0; k < MLP; ++k) acc ^= v[k].x + v[k].w + (BOTH ? w[k].y : 0u);
    }
    out[gid] = acc;
}

int main(int argc, char** argv) {
    const double mb = argc > 1 ? atof(argv[1]) : 150;   // table size in MB (fractions allowed: 0.015625 = 16 KB, L1-resident)
    const unsigned stride = argc > 2 ? atoi(argv[2]) : 64;       // bytes between slots
    const int blocks_per_cu = argc > 3 ? atoi(argv[3]) : 6;
    const int iters = argc > 4 ? atoi(argv[4]) : 256;
    const size_t bytes = (size_t)(mb * 1048576.0);
    uint4* table; unsigned* out;
    hipMalloc((void**)&table, bytes);
    hipMemset(table, 1, bytes);
    const unsigned grid = 256 * blocks_per_cu;
    hipMalloc((void**)&out, (size_t)grid * 256 * 4);
    const unsigned n_slots = (unsigned)(bytes / stride);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mlp : {1, 2, 4, 8}) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            switch (mlp) {
                case 1: gather<1><<<grid, 256>>>(table, n_slots, stride / 16, iters * 8, out); break;
                case 2: gather<2><<<grid, 256>>>(table, n_slots, stride / 16, iters * 4, out); break;
                case 4: gather<4><<<grid, 256>>>(table, n_slots, stride / 16, iters * 2, out); break;
                default: gather<8><<<grid, 256>>>(table, n_slots, stride / 16, iters, out); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        const double lines = (double)grid * 256 * iters * 8;
        printf("table %g MB stride %u B, %d blocks/CU, %d loads in flight per lane: %.3f ms, %.1f G lines/s, %.2f TB/s at 64 B/line\n",
               mb, stride, blocks_per_cu, mlp, best, lines / best / 1e6, lines * 64 / best / 1e9);
    }
    if (stride == 128) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            gather<4, true><<<grid, 256>>>(table, n_slots, stride / 16, iters * 2, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        const double lines = (double)grid * 256 * iters * 8;
        printf("table %g MB, both 64-byte halves of random 128-byte lines: %.3f ms, %.1f G 128-byte lines/s, %.2f TB/s at 128 B/line\n",
               mb, best, lines / best / 1e6, lines * 128 / best / 1e9);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("error\n"); return 1; }
    return 0;
}
