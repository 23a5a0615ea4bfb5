// Micro-benchmark: what does a divergent per-lane load cost the texture-address / L1 pipeline?
// Every ACTIVE lane loads W bytes from a random 128-byte line of a table; only `active` of the 64
// lanes of a wave take part (the first `active` lanes, or every (64/active)-th lane when spread=1).
// Reported: wave-level load instructions per second and lane-loads per second.  If the cost of an
// instruction were proportional to its active lanes, lane-loads/s would not depend on `active`.
// build: hipcc -O3 --offload-arch=gfx950 -o build/gather_mask tools/micro/gather_mask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <int W>
__global__ void __launch_bounds__(256) gather(const unsigned* __restrict__ table, unsigned n_lines, int iters,
                                              int active, int spread, unsigned* __restrict__ out) {
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    const unsigned lane = threadIdx.x & 63u;
    const bool on = spread ? (lane % (64u / (unsigned)active)) == 0u : lane < (unsigned)active;
    unsigned acc = 0, h = hash32(gid + 1u);
    if (on) {
        for (int it = 0; it < iters; ++it) {
            h = hash32(h + 0x9E3779B9u + acc);   // dependent on the previous load: one load in flight per lane
            const unsigned* p = table + (size_t)(h % n_lines) * 32u;
            if (W == 4) acc ^= *p;
            else if (W == 8) { const uint2 v = *reinterpret_cast<const uint2*>(p); acc ^= v.x + v.y; }
            else { const uint4 v = *reinterpret_cast<const uint4*>(p); acc ^= v.x + v.w; }
        }
    }
    out[gid] = acc;
}

int main(int argc, char** argv) {
    const double mb = argc > 1 ? atof(argv[1]) : 150;
    const int blocks_per_cu = argc > 2 ? atoi(argv[2]) : 6;
    const int iters = argc > 3 ? atoi(argv[3]) : 512;
    const size_t bytes = (size_t)(mb * 1048576.0);
    unsigned* table; unsigned* out;
    if (hipMalloc((void**)&table, bytes) != hipSuccess) return 1;
    hipMemset(table, 0, bytes);
    const unsigned grid = 256 * blocks_per_cu;
    if (hipMalloc((void**)&out, (size_t)grid * 256 * 4) != hipSuccess) return 1;
    const unsigned n_lines = (unsigned)(bytes / 128);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w : {4, 8, 16})
        for (int spread : {0, 1})
            for (int active : {64, 32, 16, 8, 4}) {
                if (spread && active == 64) continue;
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0);
                    if (w == 4) gather<4><<<grid, 256>>>(table, n_lines, iters, active, spread, out);
                    else if (w == 8) gather<8><<<grid, 256>>>(table, n_lines, iters, active, spread, out);
                    else gather<16><<<grid, 256>>>(table, n_lines, iters, active, spread, out);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep > 0 && ms < best) best = ms;
                }
                const double instr = (double)grid * 4 * iters;
                printf("table %g MB, %2d B/lane, %2d lanes %s: %.3f ms, %.2f G wave-loads/s, %.1f G lane-loads/s\n", mb, w,
                       active, spread ? "spread " : "packed ", best, instr / best / 1e6, instr * active / best / 1e6);
            }
    if (hipDeviceSynchronize() != hipSuccess) { printf("error\n"); return 1; }
    return 0;
}
