// scripts/hip/valu_issue.hip -- what one wave64 VALU instruction costs on a gfx950 SIMD, by the number of waves resident on it.
// Settles the question behind DESIGN.md section 5 (is k_gn_accumulate VALU-issue bound?): /opt/skills/guides/MI355X_MICROARCH.md
// says SIMD-32, 2 cycles per wave64 v_fma_f32 with several waves, 4 for one wave alone; round 3's DESIGN priced 4 at any occupancy.
//
// Every wave runs 2048 x 32 INDEPENDENT instructions of one kind (16 destination registers, each written twice per round, one asm
// block per round) between two s_memtime stamps; blocks of 256 x w threads put w waves on each of a CU's four SIMDs (w <= 4: one block
// per CU; w = 6 / 8: two blocks of 3 / 4 per CU, grid = 2 x CUs).  The arbiter serves the oldest waves first, so a single wave's own
// elapsed time hides the starved ones: the figure reported is the BLOCK's elapsed time (first start .. last end) / (count x w) = cycles
// per wave-instruction per SIMD, i.e. the issue cost that bounds a kernel keeping w waves busy; the shader clock comes from
// s_memtime / s_memrealtime (100 MHz), and the same figure is derived from the kernel's wall time as a cross-check.
//   hipcc --offload-arch=gfx950 -O2 scripts/hip/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue > profiles/r04_valu_issue.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

enum Kind { K_FMA, K_FMAC, K_PKFMA, K_ADD, K_MUL, K_PKADD, K_CNDMASK, K_CMP, K_CMPX_S, K_MINMAX, K_RCP, K_CVT_I, K_CVT_F64, K_ADD_F64, K_FMA_F64, K_DPP_MOV, K_ADD_U32, K_LSHL_ADD, K_FMA_DEP, K_CND_DIFF, K_CND_SGPR, K_CMP_CND, K_AND, K_BFI, K_MOV, K_MAX, K_LDS_ADD64_SAME, K_LDS_ADD64_DISTINCT, K_LDS_READ_B32, K_COUNT };
static const char* kNames[K_COUNT] = {"v_fma_f32 (3 VGPR sources, 16 independent chains)", "v_fmac_f32 (VOP2: 2 sources + accumulate)", "v_pk_fma_f32 (two FMAs per lane)", "v_add_f32", "v_mul_f32", "v_pk_add_f32", "v_cndmask_b32 (vcc mask)", "v_cmp_lt_f32 (writes vcc)", "v_cmp_lt_f32 (writes an SGPR pair)", "v_min_f32", "v_rcp_f32 (transcendental)", "v_cvt_i32_f32", "v_cvt_f64_f32", "v_add_f64", "v_fma_f64", "v_mov_b32 dpp row_shr:1", "v_add_u32", "v_lshl_add_u32", "v_fma_f32, ONE dependent chain (latency)", "v_cndmask_b32, destination differs from both sources", "v_cndmask_b32_e64, mask in an SGPR pair", "v_cmp_lt_f32 vcc + v_cndmask_b32 pairs (16 pairs per round)", "v_and_b32", "v_bfi_b32", "v_mov_b32", "v_max_f32", "ds_add_u64, 16 neighbouring lanes per address", "ds_add_u64, 64 distinct addresses", "ds_read_b32 (dependent address chain: latency)"};

// block-wide elapsed time: first stamp after the barrier .. last wave's end stamp (LDS min / max), so that an unfair arbiter (oldest wave first) cannot hide starved waves
template <int KIND>
__global__ void k_issue(unsigned long long* out, int reps, float seed) {
    __shared__ unsigned long long lds[1024];
    __shared__ unsigned long long tmin, tmax, rmin, rmax;
    float v[16]; float2 p[16]; double d[16];
    for (int i = 0; i < 16; i++) { v[i] = seed + (float)i; p[i] = make_float2(seed, seed + 1.f); d[i] = (double)seed + i; }
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = (unsigned long long)((i * 8 + 8) & 8191);
    if (threadIdx.x == 0) { tmin = ~0ull; tmax = 0ull; rmin = ~0ull; rmax = 0ull; }
    const float a = seed * 0.5f, b = seed * 0.25f;
    const float2 pa = make_float2(a, b);
    const double da = (double)a;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t laddr = (uint32_t)(size_t)(KIND == K_LDS_ADD64_SAME ? ((threadIdx.x >> 6) * 64 + (lane >> 4)) : threadIdx.x) * 8u;
    uint32_t chase = (threadIdx.x * 8u) & 8191u;
    unsigned long long one = 1ull;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    // one asm block of 32 instructions per round (the compiler puts nothing in between): operand %i = register i of the 16, %16 / %17 the two sources
#define R16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define O16(A) "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]), "+v"(A[8]), "+v"(A[9]), "+v"(A[10]), "+v"(A[11]), "+v"(A[12]), "+v"(A[13]), "+v"(A[14]), "+v"(A[15])
#define I_K_FMA(i) "v_fma_f32 %" #i ", %16, %17, %" #i "\n"
#define I_K_FMAC(i) "v_fmac_f32 %" #i ", %16, %17\n"
#define I_K_PKFMA(i) "v_pk_fma_f32 %" #i ", %16, %16, %" #i "\n"
#define I_K_ADD(i) "v_add_f32 %" #i ", %16, %" #i "\n"
#define I_K_MUL(i) "v_mul_f32 %" #i ", %16, %" #i "\n"
#define I_K_PKADD(i) "v_pk_add_f32 %" #i ", %16, %" #i "\n"
#define I_K_CNDMASK(i) "v_cndmask_b32 %" #i ", %" #i ", %16, vcc\n"
#define I_K_CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %16\n"
#define I_K_CMPX_S(i) "v_cmp_lt_f32 s[20:21], %" #i ", %16\n"
#define I_K_MINMAX(i) "v_min_f32 %" #i ", %16, %" #i "\n"
#define I_K_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define I_K_CVT_I(i) "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define I_K_CVT_F64(i) "v_cvt_f64_f32 %" #i ", %16\n"
#define I_K_ADD_F64(i) "v_add_f64 %" #i ", %" #i ", %16\n"
#define I_K_FMA_F64(i) "v_fma_f64 %" #i ", %16, %16, %" #i "\n"
#define I_K_DPP_MOV(i) "v_mov_b32_dpp %" #i ", %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_K_ADD_U32(i) "v_add_u32 %" #i ", %16, %" #i "\n"
#define I_K_LSHL_ADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %16\n"
#define I_K_FMA_DEP(i) "v_fma_f32 %0, %16, %0, %17\n"
#define I_K_CND_DIFF(i) "v_cndmask_b32 %" #i ", %16, %17, vcc\n"
#define I_K_CND_SGPR(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %16, s[20:21]\n"
#define I_K_AND(i) "v_and_b32 %" #i ", %16, %" #i "\n"
#define I_K_BFI(i) "v_bfi_b32 %" #i ", %16, %17, %" #i "\n"
#define I_K_MOV(i) "v_mov_b32 %" #i ", %16\n"
#define I_K_MAX(i) "v_max_f32 %" #i ", %16, %" #i "\n"
#define I_K_CMP_CND(i) "v_cmp_lt_f32 vcc, %" #i ", %16\nv_cndmask_b32 %" #i ", %" #i ", %17, vcc\n"
    for (int r = 0; r < reps; r++) {
        if (KIND == K_FMA) asm volatile(R16(I_K_FMA) R16(I_K_FMA) : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_FMAC) asm volatile(R16(I_K_FMAC) R16(I_K_FMAC) : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_PKFMA) asm volatile(R16(I_K_PKFMA) R16(I_K_PKFMA) : O16(p) : "v"(pa), "v"(pa) : "vcc", "s20", "s21");
        else if (KIND == K_ADD) asm volatile(R16(I_K_ADD) R16(I_K_ADD) : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_MUL) asm volatile(R16(I_K_MUL) R16(I_K_MUL) : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_PKADD) asm volatile(R16(I_K_PKADD) R16(I_K_PKADD) : O16(p) : "v"(pa), "v"(pa) : "vcc", "s20", "s21");
        else if (KIND == K_CNDMASK) asm volatile(R16(I_K_CNDMASK) R16(I_K_CNDMASK) : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_CMP) asm volatile(R16(I_K_CMP) R16(I_K_CMP) : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_CMPX_S) asm volatile(R16(I_K_CMPX_S) R16(I_K_CMPX_S) : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_MINMAX) asm volatile(R16(I_K_MINMAX) R16(I_K_MINMAX) : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_RCP) asm volatile(R16(I_K_RCP) R16(I_K_RCP) : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_CVT_I) asm volatile(R16(I_K_CVT_I) R16(I_K_CVT_I) : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_CVT_F64) asm volatile(R16(I_K_CVT_F64) R16(I_K_CVT_F64) : O16(d) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_ADD_F64) asm volatile(R16(I_K_ADD_F64) R16(I_K_ADD_F64) : O16(d) : "v"(da), "v"(da) : "vcc", "s20", "s21");
        else if (KIND == K_FMA_F64) asm volatile(R16(I_K_FMA_F64) R16(I_K_FMA_F64) : O16(d) : "v"(da), "v"(da) : "vcc", "s20", "s21");
        else if (KIND == K_DPP_MOV) asm volatile(R16(I_K_DPP_MOV) R16(I_K_DPP_MOV) : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_ADD_U32) asm volatile(R16(I_K_ADD_U32) R16(I_K_ADD_U32) : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_LSHL_ADD) asm volatile(R16(I_K_LSHL_ADD) R16(I_K_LSHL_ADD) : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_FMA_DEP) asm volatile(R16(I_K_FMA_DEP) R16(I_K_FMA_DEP) : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_CND_DIFF) asm volatile(R16(I_K_CND_DIFF) R16(I_K_CND_DIFF) : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_CND_SGPR) asm volatile(R16(I_K_CND_SGPR) R16(I_K_CND_SGPR) : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_CMP_CND) asm volatile(R16(I_K_CMP_CND)  : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_AND) asm volatile(R16(I_K_AND) R16(I_K_AND) : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_BFI) asm volatile(R16(I_K_BFI) R16(I_K_BFI) : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_MOV) asm volatile(R16(I_K_MOV) R16(I_K_MOV) : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_MAX) asm volatile(R16(I_K_MAX) R16(I_K_MAX) : O16(v) : "v"(a), "v"(b) : "vcc", "s20", "s21");
        else if (KIND == K_LDS_READ_B32) {
#pragma unroll
            for (int i = 0; i < 32; i++) asm volatile("ds_read_b32 %0, %0\ns_waitcnt lgkmcnt(0)" : "+v"(chase) : : "memory");
        } else {
#pragma unroll
            for (int i = 0; i < 32; i++) asm volatile("ds_add_u64 %0, %1" : : "v"(laddr), "v"(one) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = (float)chase; for (int i = 0; i < 16; i++) s += v[i] + p[i].x + p[i].y + (float)d[i];
    if (s == 1.2345e-30f) out[0] = 1;                                          // keeps the values alive
    if ((threadIdx.x & 63) == 0) { atomicMin(&tmin, t0); atomicMax(&tmax, t1); atomicMin(&rmin, r0); atomicMax(&rmax, r1); }
    __syncthreads();
    if (threadIdx.x == 0) { out[1 + 2 * blockIdx.x] = tmax - tmin; out[2 + 2 * blockIdx.x] = rmax - rmin; }
}

template <int KIND> void run(unsigned long long* d_out, int n_cu, FILE* f) {
    const int reps = 2048;
    fprintf(f, "%s\n", kNames[KIND]);
    fprintf(f, "  waves/SIMD   cycles per wave-instr per SIMD (block elapsed / (count x w), median over blocks)   shader clock GHz   kernel us   same from kernel time at that clock\n");
    const int ws[] = {1, 2, 3, 4, 6, 8};
    for (int w : ws) {
        const int blocks_per_cu = w > 4 ? 2 : 1, threads = 256 * (w > 4 ? w / 2 : w), grid = n_cu * blocks_per_cu;
        hipMemset(d_out, 0, sizeof(unsigned long long) * (1 + 2 * grid));
        k_issue<KIND><<<grid, threads>>>(d_out, 16, 1.0f);                     // warm-up (clocks, code)
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k_issue<KIND><<<grid, threads>>>(d_out, reps, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(1 + 2 * grid);
        hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
        std::vector<double> cyc, ghz;
        for (int b = 0; b < grid; b++) { cyc.push_back((double)h[1 + 2 * b]); ghz.push_back((double)h[1 + 2 * b] / ((double)h[2 + 2 * b] * 10.0)); }   // s_memrealtime: 100 MHz
        std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
        const double n = (double)reps * 32.0, wres = (double)w / blocks_per_cu;   // waves per SIMD of ONE block; with two blocks per CU the blocks overlap in time
        const double per_simd = cyc[cyc.size() / 2] / (n * wres) / blocks_per_cu;
        const double clk = ghz[ghz.size() / 2];
        fprintf(f, "  %6d       %10.2f                                                                             %6.3f          %8.1f     %8.2f\n", w, per_simd, clk, ms * 1e3, ms * 1e-3 * clk * 1e9 / (n * w));
    }
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount;
    unsigned long long* d_out; hipMalloc(&d_out, sizeof(unsigned long long) * (1 + n_cu * 2 * 2 + 64));
    FILE* f = stdout;
    fprintf(f, "VALU issue cost on %s (%d CUs), s_memtime ticks (= shader cycles), 2048 x 32 instructions per wave, every CU busy\n", p.gcnArchName, n_cu);
    fprintf(f, "w waves per SIMD = blocks of 256 x w threads, one per CU (w <= 4) or two of 256 x w/2 (w = 6, 8)\n\n");
    run<K_FMA>(d_out, n_cu, f); run<K_FMAC>(d_out, n_cu, f); run<K_PKFMA>(d_out, n_cu, f); run<K_ADD>(d_out, n_cu, f); run<K_MUL>(d_out, n_cu, f); run<K_PKADD>(d_out, n_cu, f); run<K_CNDMASK>(d_out, n_cu, f); run<K_CMP>(d_out, n_cu, f); run<K_CMPX_S>(d_out, n_cu, f); run<K_MINMAX>(d_out, n_cu, f); run<K_RCP>(d_out, n_cu, f); run<K_CVT_I>(d_out, n_cu, f); run<K_CVT_F64>(d_out, n_cu, f); run<K_ADD_F64>(d_out, n_cu, f); run<K_FMA_F64>(d_out, n_cu, f); run<K_DPP_MOV>(d_out, n_cu, f); run<K_ADD_U32>(d_out, n_cu, f); run<K_LSHL_ADD>(d_out, n_cu, f); run<K_FMA_DEP>(d_out, n_cu, f); run<K_CND_DIFF>(d_out, n_cu, f); run<K_CND_SGPR>(d_out, n_cu, f); run<K_CMP_CND>(d_out, n_cu, f); run<K_AND>(d_out, n_cu, f); run<K_BFI>(d_out, n_cu, f); run<K_MOV>(d_out, n_cu, f); run<K_MAX>(d_out, n_cu, f);
    run<K_LDS_ADD64_SAME>(d_out, n_cu, f); run<K_LDS_ADD64_DISTINCT>(d_out, n_cu, f); run<K_LDS_READ_B32>(d_out, n_cu, f);
    hipFree(d_out);
    return 0;
}
