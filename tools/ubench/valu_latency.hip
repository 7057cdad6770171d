// valu_latency.hip -- LATENCY of dependent gfx950 vector instructions: one wave, one chain (every instruction reads the result of
// the one before it), the way the serial loops of the SILK synthesis run (LPC recurrence, all-pass up-sampler).  valu_rate*.hip
// measures issue RATE with independent chains; a serial loop at four waves per SIMD is paced by the chain instead.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_latency valu_latency.hip ; run on the GPU box.  Prints ns per instruction of one
// resident wave (x clock = cycles), and with W such waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define ITER 20000
#define REP16(x) x x x x x x x x x x x x x x x x
#define KERNEL(name, line)                                                                        \
    __global__ void __launch_bounds__(64) name(int *out, int b, int c) {                         \
        int a = threadIdx.x + b;                                                                  \
        for (int it = 0; it < ITER; it++) { REP16(asm volatile(line "\n" : "+v"(a) : "v"(b), "v"(c));) } \
        out[blockIdx.x * 64 + threadIdx.x] = a;                                                   \
    }
KERNEL(k_add, "v_add_u32 %0, %0, %1")
KERNEL(k_mulhi, "v_mul_hi_i32 %0, %0, %1")
KERNEL(k_mul24, "v_mul_i32_i24 %0, %0, %1")
KERNEL(k_mad24, "v_mad_i32_i24 %0, %0, %1, %2")
KERNEL(k_med3, "v_med3_i32 %0, %0, %1, %2")
KERNEL(k_addsat, "v_add_i32 %0, %0, %1 clamp")
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL(k_shl, "v_lshlrev_b32 %0, 4, %0")
KERNEL(k_dppmov, "s_nop 1\n v_mov_b32_dpp %0, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf bound_ctrl:1")
KERNEL(k_dppadd, "s_nop 1\n v_add_u32_dpp %0, %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
KERNEL(k_dot2, "v_dot2_i32_i16 %0, %0, %1, %2")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
__global__ void __launch_bounds__(64) k_mad64(int *out, int b, int c) {
    long long a = threadIdx.x + b;
    int lo = threadIdx.x + b;
    for (int it = 0; it < ITER; it++) {
        REP16(asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, 0" : "=v"(a) : "v"(lo), "v"(b) : "vcc"); lo = (int)a;)
    }
    out[blockIdx.x * 64 + threadIdx.x] = (int)a;
}
// the LPC recurrence's loop-carried chain as the kernel has it: mul_hi -> add (dpp) -> add -> med3 -> shl -> add clamp -> mov dpp
__global__ void __launch_bounds__(64) k_lpc_chain(int *out, int b, int c) {
    int sn = threadIdx.x + b, R = c;
    for (int it = 0; it < ITER; it++) {
        REP16(asm volatile("v_mul_hi_i32 %1, %0, %2\n s_nop 1\n v_add_u32_dpp %1, %1, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32 %0, %3, %1\n"
                           "v_med3_i32 %0, %0, %2, %3\n v_lshlrev_b32 %0, 4, %0\n v_add_i32 %0, %0, %2 clamp\n s_nop 1\n"
                           "v_mov_b32_dpp %0, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" : "+v"(sn), "+v"(R) : "v"(b), "v"(c));)
    }
    out[blockIdx.x * 64 + threadIdx.x] = sn + R;
}
typedef void (*kfn)(int *, int, int);
int main() {
    int *d;
    hipMalloc(&d, 1024 * 8 * 64 * 4);
    struct { const char *name; kfn f; int per; } ks[] = {
        {"v_add_u32", k_add, 1}, {"v_mul_hi_i32", k_mulhi, 1}, {"v_mul_i32_i24", k_mul24, 1}, {"v_mad_i32_i24", k_mad24, 1}, {"v_mad_i64_i32 (lo -> next)", k_mad64, 1},
        {"v_med3_i32", k_med3, 1}, {"v_add_i32 clamp", k_addsat, 1}, {"v_add3_u32", k_add3, 1}, {"v_lshlrev_b32", k_shl, 1}, {"v_dot2_i32_i16", k_dot2, 1}, {"v_perm_b32", k_perm, 1},
        {"s_nop 1 + v_mov_b32_dpp", k_dppmov, 1}, {"s_nop 1 + v_add_u32_dpp", k_dppadd, 1}, {"LPC sample (7 dependent instructions + 2 s_nop)", k_lpc_chain, 1}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-48s %10s %10s %10s %10s   (ns per chain step; x 2.4 = cycles at 2.4 GHz)\n", "dependent chain of", "1 wave", "2 / SIMD", "4 / SIMD", "8 / SIMD");
    for (auto &k : ks) {
        printf("%-48s", k.name);
        for (int w : {0, 2, 4, 8}) {
            const int grid = w == 0 ? 1 : 1024 * w;
            hipLaunchKernelGGL(k.f, dim3(grid), dim3(64), 0, 0, d, 3, 5);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k.f, dim3(grid), dim3(64), 0, 0, d, 3, 5);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf(" %10.2f", ms * 1e6 / (16.0 * ITER));
        }
        printf("\n");
    }
    return 0;
}
