// Diagnostic (not part of the product): vector-instruction issue rate of one SIMD of gfx950 as a function of the
// number of resident waves, for the instruction kinds the solve / observation kernels are made of.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
// Grid: 256 CUs x W workgroups of 256 lanes (one wave per SIMD each), every wave runs REP x 32 independent
// instructions of one kind; cycles per instruction per SIMD = time * clock / (W * REP * 32).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP 4000

#define BODY8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)
#define KERNEL(NAME, DECL, INS, SINK)                                                             \
    __global__ __launch_bounds__(256) void NAME(float* out, int rep, float seed) {                \
        DECL                                                                                      \
        for (int r = 0; r < rep; ++r) {                                                           \
            BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS)                                           \
        }                                                                                         \
        SINK                                                                                      \
    }

#define DECL_F float v0 = seed + threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7, c = seed * 0.5f;
#define SINK_F if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 == 12345.0f) out[threadIdx.x] = v0;
#define INS_ADD(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v##k) : "v"(c));
#define INS_FMA(k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v##k) : "v"(c));
#define INS_MED3(k) asm volatile("v_med3_u32 %0, %0, %1, %1" : "+v"(v##k) : "v"(c));
#define INS_CNDMASK(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v##k) : "v"(c));
#define INS_RCP(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(v##k));
#define INS_SQRT(k) asm volatile("v_sqrt_f32 %0, %0" : "+v"(v##k));
#define INS_DIVFIX(k) asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(v##k) : "v"(c));
#define INS_DIVSCALE(k) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %1" : "+v"(v##k) : "v"(c) : "vcc", "scc");
#define INS_DIVFMAS(k) asm volatile("v_div_fmas_f32 %0, %0, %1, %1" : "+v"(v##k) : "v"(c));
#define INS_CMP(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(v##k), "v"(c) : "vcc");
#define INS_CMPS(k) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(v##k), "v"(c) : "s20", "s21");
#define INS_DPP(k) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v##k));

#define INS_CNDS(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(v##k) : "v"(c));
#define INS_CMPCND(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v##k) : "v"(c) : "vcc", "scc");
#define INS_MOV(k) asm volatile("v_mov_b32 %0, %1" : "=v"(v##k) : "v"(c));
#define INS_MULE32(k) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(v##k) : "v"(c));
#define INS_MULE64(k) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(v##k) : "v"(c));
#define INS_FMAC(k) asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(v##k) : "v"(c));
#define INS_MAX(k) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(v##k) : "v"(c));
#define INS_AND(k) asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(v##k) : "v"(c));
#define INS_MINU(k) asm volatile("v_min_u32_e32 %0, %0, %1" : "+v"(v##k) : "v"(c));
#define INS_LSHL(k) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(v##k));
#define INS_CVT(k) asm volatile("v_cvt_u32_f32_e32 %0, %0" : "+v"(v##k));
#define INS_ADDU(k) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(v##k) : "v"(c));
#define INS_SUBABS(k) asm volatile("v_sub_f32_e64 %0, |%0|, %1" : "+v"(v##k) : "v"(c));
#define INS_ADDSG(k) asm volatile("v_add_f32_e32 %0, s20, %0" : "+v"(v##k));
#define INS_READLANE(k) asm volatile("v_readlane_b32 s22, %0, 3" : : "v"(v##k) : "s22");
#define DECL_CND DECL_F asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b64 vcc, 0x3333" ::: "s20", "s21", "vcc");
#define INS_PA(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_and_b64 vcc, s[20:21], vcc\n s_nop 0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v##k) : "v"(c) : "vcc", "scc");
#define INS_PB(k) asm volatile("v_cmp_lt_f32 s[22:23], %0, %1\n s_and_b64 s[22:23], s[20:21], s[22:23]\n s_nop 0\n v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(v##k) : "v"(c) : "s22", "s23", "scc");
#define INS_PC(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(v##k) : "v"(c) : "vcc", "scc");
#define INS_PD(k) asm volatile("s_and_b64 vcc, s[20:21], exec\n s_nop 0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v##k) : "v"(c) : "vcc", "scc");
#define INS_PE(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_add_f32 %0, %0, %1\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(v##k) : "v"(c) : "vcc", "scc");
#define INS_PF(k) asm volatile("v_cmp_lt_f32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc" : "+v"(v##k), "+v"(w##k) : "v"(c) : "vcc", "scc");
#define INS_PG(k) asm volatile("v_cmp_lt_f32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %3, %3, %2, vcc\n v_cndmask_b32 %4, %4, %2, vcc" : "+v"(v##k), "+v"(w##k), "+v"(x##k), "+v"(y##k) : "v"(c) : "vcc", "scc");
#define INS_PH(k) asm volatile("v_cmp_lt_f32 s[22:23], %0, %2\n v_cndmask_b32_e64 %0, %0, %2, s[22:23]\n v_cndmask_b32_e64 %1, %1, %2, s[22:23]\n v_cndmask_b32_e64 %3, %3, %2, s[22:23]\n v_cndmask_b32_e64 %4, %4, %2, s[22:23]" : "+v"(v##k), "+v"(w##k), "+v"(x##k), "+v"(y##k) : "v"(c) : "s22", "s23", "scc");
#define DECL_W DECL_F float w0 = v0 * 2, w1 = v1 * 2, w2 = v2 * 2, w3 = v3 * 2, w4 = v4 * 2, w5 = v5 * 2, w6 = v6 * 2, w7 = v7 * 2; \
    float x0 = v0 * 3, x1 = v1 * 3, x2 = v2 * 3, x3 = v3 * 3, x4 = v4 * 3, x5 = v5 * 3, x6 = v6 * 3, x7 = v7 * 3; \
    float y0 = v0 * 5, y1 = v1 * 5, y2 = v2 * 5, y3 = v3 * 5, y4 = v4 * 5, y5 = v5 * 5, y6 = v6 * 5, y7 = v7 * 5;
#define SINK_W if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + w0 + w1 + w2 + w3 + w4 + w5 + w6 + w7 + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7 == 12345.0f) out[threadIdx.x] = v0;
#define DECL_D double v0 = seed + threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7, c = seed * 0.5;
#define SINK_D if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 == 12345.0) out[threadIdx.x] = (float)v0;
#define INS_MIN64(k) asm volatile("v_min_f64 %0, %0, %1" : "+v"(v##k) : "v"(c));
#define INS_MUL64(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v##k) : "v"(c));
#define INS_FMA64(k) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v##k) : "v"(c));
#define INS_PKMUL(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v##k) : "v"(c));
#define INS_PKFMA(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v##k) : "v"(c));

KERNEL(k_add, DECL_F, INS_ADD, SINK_F)
KERNEL(k_fma, DECL_F, INS_FMA, SINK_F)
KERNEL(k_med3, DECL_F, INS_MED3, SINK_F)
KERNEL(k_cndmask, DECL_F, INS_CNDMASK, SINK_F)
KERNEL(k_cnd_init, DECL_CND, INS_CNDMASK, SINK_F)
KERNEL(k_cnds, DECL_CND, INS_CNDS, SINK_F)
KERNEL(k_cmpcnd, DECL_F, INS_CMPCND, SINK_F)
KERNEL(k_mov, DECL_F, INS_MOV, SINK_F)
KERNEL(k_mule32, DECL_F, INS_MULE32, SINK_F)
KERNEL(k_mule64, DECL_F, INS_MULE64, SINK_F)
KERNEL(k_fmac, DECL_F, INS_FMAC, SINK_F)
KERNEL(k_max, DECL_F, INS_MAX, SINK_F)
KERNEL(k_and, DECL_F, INS_AND, SINK_F)
KERNEL(k_minu, DECL_F, INS_MINU, SINK_F)
KERNEL(k_lshl, DECL_F, INS_LSHL, SINK_F)
KERNEL(k_cvt, DECL_F, INS_CVT, SINK_F)
KERNEL(k_addu, DECL_F, INS_ADDU, SINK_F)
KERNEL(k_subabs, DECL_F, INS_SUBABS, SINK_F)
KERNEL(k_addsg, DECL_CND, INS_ADDSG, SINK_F)
KERNEL(k_readlane, DECL_F, INS_READLANE, SINK_F)
KERNEL(k_pa, DECL_CND, INS_PA, SINK_F)
KERNEL(k_pb, DECL_CND, INS_PB, SINK_F)
KERNEL(k_pc, DECL_CND, INS_PC, SINK_F)
KERNEL(k_pd, DECL_CND, INS_PD, SINK_F)
KERNEL(k_pe, DECL_CND, INS_PE, SINK_F)
KERNEL(k_pf, DECL_W, INS_PF, SINK_W)
KERNEL(k_pg, DECL_W, INS_PG, SINK_W)
KERNEL(k_ph, DECL_W, INS_PH, SINK_W)
KERNEL(k_rcp, DECL_F, INS_RCP, SINK_F)
KERNEL(k_sqrt, DECL_F, INS_SQRT, SINK_F)
KERNEL(k_divfix, DECL_F, INS_DIVFIX, SINK_F)
KERNEL(k_divscale, DECL_F, INS_DIVSCALE, SINK_F)
KERNEL(k_divfmas, DECL_F, INS_DIVFMAS, SINK_F)
KERNEL(k_cmp_vcc, DECL_F, INS_CMP, SINK_F)
KERNEL(k_cmp_sgpr, DECL_F, INS_CMPS, SINK_F)
KERNEL(k_dpp, DECL_F, INS_DPP, SINK_F)
KERNEL(k_min64, DECL_D, INS_MIN64, SINK_D)
KERNEL(k_mul64, DECL_D, INS_MUL64, SINK_D)
KERNEL(k_fma64, DECL_D, INS_FMA64, SINK_D)
KERNEL(k_pkmul, DECL_D, INS_PKMUL, SINK_D)
KERNEL(k_pkfma, DECL_D, INS_PKFMA, SINK_D)

// a dependent chain: one accumulator (latency per instruction when the wave is alone)
__global__ __launch_bounds__(256) void k_add_chain(float* out, int rep, float seed) {
    float v0 = seed + threadIdx.x, c = seed * 0.5f;
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int k = 0; k < 32; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v0) : "v"(c));
    }
    if (v0 == 12345.0f) out[threadIdx.x] = v0;
}
// LDS broadcast read + use, like the candidate loop of the neighbour scan
__global__ __launch_bounds__(256) void k_lds_bcast(float* out, int rep, float seed) {
    __shared__ float s[1024];
    for (int t = threadIdx.x; t < 1024; t += 256) s[t] = seed + t;
    __syncthreads();
    float acc = 0.0f;
    for (int r = 0; r < rep; ++r) {
#pragma unroll
        for (int k = 0; k < 32; ++k) acc += s[(r + k) & 1023];
    }
    if (acc == 12345.0f) out[threadIdx.x] = acc;
}

typedef void (*kern_t)(float*, int, float);

int main() {
    float* d;
    hipMalloc(&d, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct { const char* name; kern_t k; } ks[] = {
        {"v_add_f32", k_add}, {"v_fma_f32", k_fma}, {"v_med3_u32", k_med3}, {"v_cndmask_b32", k_cndmask},
        {"v_cndmask_b32 vcc (vcc set)", k_cnd_init}, {"v_cndmask_b32_e64 sgpr", k_cnds}, {"v_cmp+v_cndmask pair (x2)", k_cmpcnd},
        {"v_mov_b32", k_mov}, {"v_mul_f32_e32", k_mule32}, {"v_mul_f32_e64", k_mule64}, {"v_fmac_f32_e32", k_fmac}, {"v_max_f32_e32", k_max},
        {"v_and_b32_e32", k_and}, {"v_min_u32_e32", k_minu}, {"v_lshlrev_b32_e32", k_lshl}, {"v_cvt_u32_f32", k_cvt}, {"v_add_u32_e32", k_addu},
        {"v_sub_f32_e64 |abs|", k_subabs}, {"v_add_f32_e32 sgpr operand", k_addsg}, {"v_readlane_b32", k_readlane},
        {"A: cmp vcc; s_and vcc; cndmask vcc", k_pa}, {"B: cmp sgpr; s_and; cndmask_e64", k_pb}, {"C: cmp; cndmask; cndmask", k_pc},
        {"D: s_and vcc; cndmask vcc", k_pd}, {"E: cmp; cnd; add; cnd", k_pe},
        {"F: cmp; cnd x; cnd y (indep)", k_pf}, {"G: cmp; 4 indep cnd vcc", k_pg}, {"H: cmp sgpr; 4 indep cnd_e64", k_ph},
        {"v_cmp_lt_f32 vcc", k_cmp_vcc}, {"v_cmp_lt_f32 sgpr", k_cmp_sgpr}, {"v_mov_dpp quad_perm", k_dpp},
        {"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt}, {"v_div_scale_f32", k_divscale}, {"v_div_fmas_f32", k_divfmas},
        {"v_div_fixup_f32", k_divfix}, {"v_min_f64", k_min64}, {"v_mul_f64", k_mul64}, {"v_fma_f64", k_fma64},
        {"v_pk_mul_f32", k_pkmul}, {"v_pk_fma_f32", k_pkfma}, {"v_add_f32 dependent chain", k_add_chain},
        {"ds_read_b32 broadcast + v_add", k_lds_bcast}};
    const int waves[] = {1, 2, 4, 8};
    printf("cycles per wave-instruction per SIMD at 2.4 GHz nominal (256 CUs x W workgroups of 4 waves; %d x 32 instructions per wave)\n", REP);
    printf("%-32s", "instruction \\ waves per SIMD");
    for (int w : waves) printf("%8d", w);
    printf("\n");
    for (auto& kk : ks) {
        printf("%-32s", kk.name);
        for (int w : waves) {
            const dim3 grid(256 * w), block(256);
            hipLaunchKernelGGL(kk.k, grid, block, 0, 0, d, 200, 1.0f);  // warm
            hipDeviceSynchronize();
            float best = 1e30f;
            for (int t = 0; t < 3; ++t) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(kk.k, grid, block, 0, 0, d, REP, 1.0f);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double cyc = best * 1e-3 * 2.4e9 / ((double)w * REP * 32);
            printf("%8.2f", cyc);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
