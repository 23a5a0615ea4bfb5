// Micro-benchmark: what does one vector-ALU wave-instruction cost on gfx950, by instruction kind, by waves per
// SIMD and by how many of its 64 lanes are enabled?  bench.py's `valu` object prices a kernel's counted
// instructions (SQ_INSTS_VALU) against a peak issue rate; this program measures that peak instead of assuming
// "one wave-instruction per 4 cycles per SIMD" (the shadow-pair kernel sustains 3.75, so 4 cannot be the floor).
// Each wave runs 32 instructions per iteration on 8 independent accumulators under a chosen EXEC mask; the whole
// chip is filled at `w/SIMD` waves per SIMD and the time gives cycles per instruction per SIMD.  The loop itself
// (5 scalar instructions per 32) adds ~0.4 cycles per instruction to every figure.
// build: hipcc -O3 --offload-arch=gfx950 -o build/valu_issue tools/micro/valu_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define R8(OP) OP("%0") OP("%1") OP("%2") OP("%3") OP("%4") OP("%5") OP("%6") OP("%7")
// 32-bit accumulators d; m, c = two more VGPRs; s[22:23] = a condition mask
#define OP_FMA(d)    "v_fma_f32 " d ", " d ", %8, %9\n"
#define OP_FMAI(d)   "v_fma_f32 " d ", %8, %9, %9\n"          /* no dependence on the accumulator */
#define OP_FMAC(d)   "v_fmac_f32 " d ", %8, %9\n"
#define OP_MUL(d)    "v_mul_f32 " d ", " d ", %8\n"
#define OP_ADD(d)    "v_add_f32 " d ", " d ", %9\n"
#define OP_SUB(d)    "v_sub_f32 " d ", " d ", %9\n"
#define OP_MAX(d)    "v_max_f32 " d ", " d ", %9\n"
#define OP_MIN(d)    "v_min_f32 " d ", " d ", %9\n"
#define OP_MED3(d)   "v_med3_f32 " d ", " d ", %9, %8\n"
#define OP_ADDU(d)   "v_add_u32 " d ", " d ", %9\n"
#define OP_SUBU(d)   "v_sub_u32 " d ", " d ", %9\n"
#define OP_AND(d)    "v_and_b32 " d ", " d ", %8\n"
#define OP_OR(d)     "v_or_b32 " d ", " d ", %8\n"
#define OP_XOR(d)    "v_xor_b32 " d ", " d ", %9\n"
#define OP_NOT(d)    "v_not_b32 " d ", " d "\n"
#define OP_SHL(d)    "v_lshlrev_b32 " d ", 1, " d "\n"
#define OP_SHR(d)    "v_lshrrev_b32 " d ", 1, " d "\n"
#define OP_BFE(d)    "v_bfe_u32 " d ", " d ", 3, 5\n"
#define OP_LSHLADD(d) "v_lshl_add_u32 " d ", " d ", 2, %9\n"
#define OP_LSHLOR(d) "v_lshl_or_b32 " d ", " d ", 2, %9\n"
#define OP_OR3(d)    "v_or3_b32 " d ", " d ", %9, %8\n"
#define OP_BFI(d)    "v_bfi_b32 " d ", %9, %8, " d "\n"
#define OP_MADU24(d) "v_mad_u32_u24 " d ", " d ", %9, %8\n"
#define OP_MULLO(d)  "v_mul_lo_u32 " d ", " d ", %8\n"
#define OP_MULHI(d)  "v_mul_hi_u32 " d ", " d ", %8\n"
#define OP_MOV(d)    "v_mov_b32 " d ", %8\n"
#define OP_CNDV(d)   "v_cndmask_b32 " d ", " d ", %8, vcc\n"
#define OP_CNDS(d)   "v_cndmask_b32 " d ", " d ", %8, s[22:23]\n"
#define OP_CNDMIX(d) "v_cndmask_b32 " d ", " d ", %8, vcc\n v_fma_f32 " d ", " d ", %8, %9\n"   /* 2 instructions */
#define OP_CMPV(d)   "v_cmp_lt_f32 vcc, " d ", %8\n"
#define OP_CMPS(d)   "v_cmp_lt_f32 s[22:23], " d ", %8\n"
#define OP_CMPU(d)   "v_cmp_eq_u32 vcc, " d ", %8\n"
#define OP_CVT(d)    "v_cvt_f32_u32 " d ", " d "\n"
#define OP_FREXP(d)  "v_frexp_exp_i32_f32 " d ", " d "\n"
#define OP_RCP(d)    "v_rcp_f32 " d ", " d "\n"
#define OP_SQRT(d)   "v_sqrt_f32 " d ", " d "\n"
#define OP_DIVSCALE(d) "v_div_scale_f32 " d ", vcc, " d ", %8, " d "\n"
#define OP_DIVFMAS(d) "v_div_fmas_f32 " d ", " d ", %8, %9\n"
#define OP_DIVFIXUP(d) "v_div_fixup_f32 " d ", " d ", %8, %9\n"
#define OP_RFL(d)    "v_readfirstlane_b32 s24, " d "\n"
// 64-bit accumulators d (register pairs); %8, %9 are pairs too
#define OP_PKMUL(d)  "v_pk_mul_f32 " d ", " d ", %8\n"
#define OP_PKADD(d)  "v_pk_add_f32 " d ", " d ", %9\n"
#define OP_PKFMA(d)  "v_pk_fma_f32 " d ", " d ", %8, %9\n"
#define OP_PKMOV(d)  "v_pk_mov_b32 " d ", %8, %9\n"
#define OP_MOV64(d)  "v_mov_b64 " d ", %8\n"
#define OP_LSHLADD64(d) "v_lshl_add_u64 " d ", " d ", 2, %9\n"
#define OP_SHL64(d)  "v_lshlrev_b64 " d ", 1, " d "\n"
#define OP_FMA64(d)  "v_fma_f64 " d ", " d ", %8, %9\n"
#define OP_ADD64(d)  "v_add_f64 " d ", " d ", %9\n"

#define BODY(OPM) asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 s[22:23], %10\n s_mov_b64 exec, %10\n" \
                               R8(OPM) R8(OPM) R8(OPM) R8(OPM) "s_mov_b64 exec, s[20:21]\n" \
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
            : "v"(m), "v"(c), "s"(mask) : "s20", "s21", "s22", "s23", "s24", "vcc")

// EXEC is restored before anything but the measured instructions runs
template <int OP>
__global__ void __launch_bounds__(256) issue(unsigned long long mask, int iters, float* __restrict__ out) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f, c = 0.001f;
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) BODY(OP_FMA);
        if (OP == 1) BODY(OP_FMAI);
        if (OP == 2) BODY(OP_FMAC);
        if (OP == 3) BODY(OP_MUL);
        if (OP == 4) BODY(OP_ADD);
        if (OP == 5) BODY(OP_SUB);
        if (OP == 6) BODY(OP_MAX);
        if (OP == 7) BODY(OP_MIN);
        if (OP == 8) BODY(OP_MED3);
        if (OP == 9) BODY(OP_ADDU);
        if (OP == 10) BODY(OP_SUBU);
        if (OP == 11) BODY(OP_AND);
        if (OP == 12) BODY(OP_OR);
        if (OP == 13) BODY(OP_XOR);
        if (OP == 14) BODY(OP_NOT);
        if (OP == 15) BODY(OP_SHL);
        if (OP == 16) BODY(OP_SHR);
        if (OP == 17) BODY(OP_BFE);
        if (OP == 18) BODY(OP_LSHLADD);
        if (OP == 19) BODY(OP_LSHLOR);
        if (OP == 20) BODY(OP_OR3);
        if (OP == 21) BODY(OP_BFI);
        if (OP == 22) BODY(OP_MADU24);
        if (OP == 23) BODY(OP_MULLO);
        if (OP == 24) BODY(OP_MULHI);
        if (OP == 25) BODY(OP_MOV);
        if (OP == 26) BODY(OP_CNDV);
        if (OP == 27) BODY(OP_CNDS);
        if (OP == 28) BODY(OP_CNDMIX);
        if (OP == 29) BODY(OP_CMPV);
        if (OP == 30) BODY(OP_CMPS);
        if (OP == 31) BODY(OP_CMPU);
        if (OP == 32) BODY(OP_CVT);
        if (OP == 33) BODY(OP_FREXP);
        if (OP == 34) BODY(OP_RCP);
        if (OP == 35) BODY(OP_SQRT);
        if (OP == 36) BODY(OP_DIVSCALE);
        if (OP == 37) BODY(OP_DIVFMAS);
        if (OP == 38) BODY(OP_DIVFIXUP);
        if (OP == 39) BODY(OP_RFL);
    }
    out[blockIdx.x * 256u + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP>
__global__ void __launch_bounds__(256) issue64(unsigned long long mask, int iters, float* __restrict__ out) {
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 0.999, c = 0.001;
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) BODY(OP_PKMUL);
        if (OP == 1) BODY(OP_PKADD);
        if (OP == 2) BODY(OP_PKFMA);
        if (OP == 3) BODY(OP_PKMOV);
        if (OP == 4) BODY(OP_MOV64);
        if (OP == 5) BODY(OP_LSHLADD64);
        if (OP == 6) BODY(OP_SHL64);
        if (OP == 7) BODY(OP_FMA64);
        if (OP == 8) BODY(OP_ADD64);
    }
    out[blockIdx.x * 256u + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}

typedef void (*kern_t)(unsigned long long, int, float*);

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 10000;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate / 1e6;
    float* out;
    if (hipMalloc((void**)&out, (size_t)cus * 8 * 256 * 4) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    struct { const char* name; kern_t k; int per; } ops[] = {
        {"v_fma_f32 d,d,m,c", issue<0>, 1}, {"v_fma_f32 d,m,c,c", issue<1>, 1}, {"v_fmac_f32", issue<2>, 1}, {"v_mul_f32", issue<3>, 1},
        {"v_add_f32", issue<4>, 1}, {"v_sub_f32", issue<5>, 1}, {"v_max_f32", issue<6>, 1}, {"v_min_f32", issue<7>, 1},
        {"v_med3_f32", issue<8>, 1}, {"v_add_u32", issue<9>, 1}, {"v_sub_u32", issue<10>, 1}, {"v_and_b32", issue<11>, 1},
        {"v_or_b32", issue<12>, 1}, {"v_xor_b32", issue<13>, 1}, {"v_not_b32", issue<14>, 1}, {"v_lshlrev_b32", issue<15>, 1},
        {"v_lshrrev_b32", issue<16>, 1}, {"v_bfe_u32", issue<17>, 1}, {"v_lshl_add_u32", issue<18>, 1}, {"v_lshl_or_b32", issue<19>, 1},
        {"v_or3_b32", issue<20>, 1}, {"v_bfi_b32", issue<21>, 1}, {"v_mad_u32_u24", issue<22>, 1}, {"v_mul_lo_u32", issue<23>, 1},
        {"v_mul_hi_u32", issue<24>, 1}, {"v_mov_b32", issue<25>, 1}, {"v_cndmask_b32 vcc", issue<26>, 1}, {"v_cndmask_b32 s[22:23]", issue<27>, 1},
        {"cndmask vcc + fma pair", issue<28>, 2}, {"v_cmp_lt_f32 vcc", issue<29>, 1}, {"v_cmp_lt_f32 s[22:23]", issue<30>, 1},
        {"v_cmp_eq_u32 vcc", issue<31>, 1}, {"v_cvt_f32_u32", issue<32>, 1}, {"v_frexp_exp_i32_f32", issue<33>, 1}, {"v_rcp_f32", issue<34>, 1},
        {"v_sqrt_f32", issue<35>, 1}, {"v_div_scale_f32", issue<36>, 1}, {"v_div_fmas_f32", issue<37>, 1}, {"v_div_fixup_f32", issue<38>, 1},
        {"v_readfirstlane_b32", issue<39>, 1},
        {"v_pk_mul_f32", issue64<0>, 1}, {"v_pk_add_f32", issue64<1>, 1}, {"v_pk_fma_f32", issue64<2>, 1}, {"v_pk_mov_b32", issue64<3>, 1},
        {"v_mov_b64", issue64<4>, 1}, {"v_lshl_add_u64", issue64<5>, 1}, {"v_lshlrev_b64", issue64<6>, 1}, {"v_fma_f64", issue64<7>, 1},
        {"v_add_f64", issue64<8>, 1}};
    struct { const char* name; unsigned long long m; } masks[] = {
        {"64", ~0ull}, {"32", 0xffffffffull}, {"16", 0xffffull}, {"4x1", 0x0001000100010001ull}, {"1", 1ull}, {"0", 0ull}};
    printf("# %d CUs, clock %.2f GHz (hipDeviceProp), %d iterations x 32 instructions per wave, 8 accumulators\n", cus, ghz, iters);
    printf("# cycles per wave-instruction per SIMD; columns = enabled lanes (4x1: one lane in each group of 16)\n");
    printf("# %-24s %6s", "instruction", "w/SIMD");
    for (auto& mk : masks) printf(" %7s", mk.name);
    printf("\n");
    for (auto& op : ops) {
        for (int wps : {1, 2, 6, 8}) {
            printf("  %-24s %6d", op.name, wps);
            for (auto& mk : masks) {
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    (void)hipEventRecord(e0);
                    op.k<<<cus * wps, 256>>>(mk.m, iters, out);
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                    if (rep > 0 && ms < best) best = ms;
                }
                printf(" %7.3f", best * 1e-3 * ghz * 1e9 / ((double)iters * 32 * op.per * wps));
            }
            printf("\n");
            fflush(stdout);
        }
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("error\n"); return 1; }
    return 0;
}
