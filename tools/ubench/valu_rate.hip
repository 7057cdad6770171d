// valu_rate.hip -- issue cost of the integer instructions the decode kernels lean on, relative to v_add_u32 (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.  Each kernel runs ITER x 32 independent
// instructions (8 chains) per wave; grids of 1024 x W one-wave workgroups = W waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <string>

#define ITER 2000
#define REP8(x) x x x x x x x x
#define BODY(asm_line)                                                                                             \
    for (int it = 0; it < ITER; it++) {                                                                            \
        REP8(asm volatile(asm_line "\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) \
    }

#define KERNEL(name, line4)                                                                                         \
    __global__ void __launch_bounds__(64) name(int *out, int b, int c) {                                          \
        int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        BODY(line4)                                                                                                 \
        out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                 \
    }

// four instructions per asm statement, on chains 0..3 then 4..7 alternately is not needed: eight statements x 4 = 32
KERNEL(k_add, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %9\n v_add_u32 %3, %3, %9")
KERNEL(k_mul24, "v_mul_i32_i24 %0, %0, %8\n v_mul_i32_i24 %1, %1, %8\n v_mul_i32_i24 %2, %2, %9\n v_mul_i32_i24 %3, %3, %9")
KERNEL(k_mad24, "v_mad_i32_i24 %0, %0, %8, %9\n v_mad_i32_i24 %1, %1, %8, %9\n v_mad_i32_i24 %2, %2, %9, %8\n v_mad_i32_i24 %3, %3, %9, %8")
KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %9\n v_mul_lo_u32 %3, %3, %9")
KERNEL(k_mulhi, "v_mul_hi_i32 %0, %0, %8\n v_mul_hi_i32 %1, %1, %8\n v_mul_hi_i32 %2, %2, %9\n v_mul_hi_i32 %3, %3, %9")
KERNEL(k_dot2, "v_dot2_i32_i16 %0, %0, %8, %9\n v_dot2_i32_i16 %1, %1, %8, %9\n v_dot2_i32_i16 %2, %2, %9, %8\n v_dot2_i32_i16 %3, %3, %9, %8")
KERNEL(k_perm, "v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %9, %8\n v_perm_b32 %3, %3, %9, %8")
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %9\n v_lshl_add_u32 %3, %3, 1, %9")
KERNEL(k_ashr, "v_ashrrev_i32 %0, 15, %0\n v_ashrrev_i32 %1, 15, %1\n v_ashrrev_i32 %2, 15, %2\n v_ashrrev_i32 %3, 15, %3")
KERNEL(k_pkmad, "v_pk_mad_i16 %0, %0, %8, %9\n v_pk_mad_i16 %1, %1, %8, %9\n v_pk_mad_i16 %2, %2, %9, %8\n v_pk_mad_i16 %3, %3, %9, %8")
KERNEL(k_pkadd, "v_pk_add_i16 %0, %0, %8\n v_pk_add_i16 %1, %1, %8\n v_pk_add_i16 %2, %2, %9\n v_pk_add_i16 %3, %3, %9")
KERNEL(k_sdwa, "v_mul_i32_i24_sdwa %0, sext(%0), %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_mul_i32_i24_sdwa %1, sext(%1), %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_mul_i32_i24_sdwa %2, sext(%2), %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_mul_i32_i24_sdwa %3, sext(%3), %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
KERNEL(k_dpp, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_adddpp, "v_add_u32_dpp %0, %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_bfe, "v_bfe_i32 %0, %0, 0, 16\n v_bfe_i32 %1, %1, 0, 16\n v_bfe_i32 %2, %2, 0, 16\n v_bfe_i32 %3, %3, 0, 16")
KERNEL(k_add3, "v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %9, %8\n v_add3_u32 %3, %3, %9, %8")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %8, 15\n v_alignbit_b32 %1, %1, %8, 15\n v_alignbit_b32 %2, %2, %9, 15\n v_alignbit_b32 %3, %3, %9, 15")
KERNEL(k_swap32, "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7")
KERNEL(k_swap16, "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7")
// 64-bit product: mad_i64_i32 writes a register pair
__global__ void __launch_bounds__(64) k_mad64(int *out, int b, int c) {
    long long a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    int x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int it = 0; it < ITER; it++) {
        REP8(asm volatile("v_mad_i64_i32 %0, vcc, %4, %8, %0\n v_mad_i64_i32 %1, vcc, %5, %8, %1\n v_mad_i64_i32 %2, vcc, %6, %9, %2\n v_mad_i64_i32 %3, vcc, %7, %9, %3\n"
                          : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(b), "v"(c) : "vcc");)
    }
    out[blockIdx.x * 64 + threadIdx.x] = (int)(a0 + a1 + a2 + a3);
}

typedef void (*kfn)(int *, int, int);
int main() {
    int *d;
    hipMalloc(&d, 1024 * 8 * 64 * 4);
    struct { const char *name; kfn f; } ks[] = {
        {"v_add_u32", k_add}, {"v_mul_i32_i24", k_mul24}, {"v_mad_i32_i24", k_mad24}, {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_i32", k_mulhi},
        {"v_mad_i64_i32", k_mad64}, {"v_dot2_i32_i16", k_dot2}, {"v_perm_b32", k_perm}, {"v_lshl_add_u32", k_lshladd}, {"v_ashrrev_i32", k_ashr},
        {"v_pk_mad_i16", k_pkmad}, {"v_pk_add_i16", k_pkadd}, {"v_mul_i32_i24_sdwa", k_sdwa}, {"v_mov_b32_dpp", k_dpp}, {"v_add_u32_dpp", k_adddpp},
        {"v_bfe_i32", k_bfe}, {"v_add3_u32", k_add3}, {"v_alignbit_b32", k_alignbit}, {"v_permlane32_swap", k_swap32}, {"v_permlane16_swap", k_swap16}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double base[9] = {0};
    for (auto &k : ks) {
        printf("%-22s", k.name);
        for (int W : {1, 2, 4, 8}) {
            hipLaunchKernelGGL(k.f, dim3(1024 * W), dim3(64), 0, 0, d, 3, 5); // warm-up
            hipDeviceSynchronize();
            float best = 1e9;
            for (int r = 0; r < 3; r++) {
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(k.f, dim3(1024 * W), dim3(64), 0, 0, d, 3, 5);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double n_inst = (std::string(k.name) == "v_mad_i64_i32" ? 32.0 : 32.0) * ITER * W; // per SIMD
            const double ns_per = best * 1e6 / n_inst;
            if (std::string(k.name) == "v_add_u32") base[W] = ns_per;
            printf("  W=%d %6.3f ns (%5.2fx add)", W, ns_per, base[W] > 0 ? ns_per / base[W] : 1.0);
        }
        printf("\n");
    }
    return 0;
}
