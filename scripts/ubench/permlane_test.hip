// permlane_test.hip -- semantics check of v_permlane16_swap / v_permlane32_swap (gfx950) as used by the
// render backward's cross-row reduction: prints what every lane holds after the swap.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
    const int lane = threadIdx.x;
    int x = 1000 + lane, y = 2000 + lane;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
    out[lane] = x; out[64 + lane] = y;
    int p = 1000 + lane, q = 2000 + lane;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(p), "+v"(q));
    out[128 + lane] = p; out[192 + lane] = q;
}
int main() {
    int* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"swap16 dst", "swap16 src", "swap32 dst", "swap32 src"};
    for (int r = 0; r < 4; r++) {
        printf("%s: rows start with", names[r]);
        for (int row = 0; row < 4; row++) printf(" %d", h[r * 64 + row * 16]);
        printf("\n");
    }
    return 0;
}
