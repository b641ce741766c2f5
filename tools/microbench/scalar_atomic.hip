// Does gfx950 execute scalar atomics (s_atomic_add_x2) with device-wide atomicity?  Every wavefront draws 100 tickets of 32;
// all tickets must be distinct multiples of 32 and the counter must end at 32 * draws.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(unsigned long long* c, unsigned long long* out, int per) {
    for (int i = 0; i < per; ++i) {
        unsigned long long r, inc = 32;
        asm volatile("s_atomic_add_x2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(c), "0"(inc) : "memory");
        if (threadIdx.x == 0) out[(size_t)blockIdx.x * per + i] = r;
    }
}
int main() {
    const int nb = 4096, per = 100;
    unsigned long long *c, *out;
    hipMalloc(&c, 8); hipMemset(c, 0, 8);
    hipMalloc(&out, (size_t)nb * per * 8);
    hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, c, out, per);
    if (hipDeviceSynchronize() != hipSuccess) { printf("FAILED: kernel error\n"); return 1; }
    std::vector<unsigned long long> h((size_t)nb * per);
    unsigned long long fin;
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(&fin, c, 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    bool ok = fin == 32ull * nb * per;
    for (size_t i = 0; i < h.size(); ++i) ok = ok && h[i] == 32ull * i;
    printf("%s: final %llu (expected %llu), first %llu last %llu\n", ok ? "OK" : "MISMATCH", fin, 32ull * nb * per, h.front(), h.back());
    return ok ? 0 : 2;
}
