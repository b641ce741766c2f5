// Streaming ceiling of the hot kernel's access pattern: NIN input rows and NOUT output rows of B doubles each (SoA), one
// point per lane, no arithmetic to speak of.  Variants: CH consecutive 256-point chunks per block (CH = 1 is the
// product kernels' mapping), and the XCD-aware block remap.  Prints ms and TB/s for each.
//   hipcc -O3 --offload-arch=gfx950 tools/soa_stream_bench.hip -o gpurun_out/soa_stream_bench && ./soa_stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int NIN, int NOUT, int CH, bool XCD>
__global__ __launch_bounds__(256) void k_stream(const double* __restrict__ in, double* __restrict__ out, long B, long nchunks) {
    long blk = blockIdx.x;
    if (XCD) {   // blocks are dealt round-robin to the 8 XCDs: give each XCD a contiguous range of chunks
        const long per = (gridDim.x + 7) / 8;
        blk = (blockIdx.x % 8) * per + blockIdx.x / 8;
        if (blk >= gridDim.x) return;
    }
    for (int c = 0; c < CH; ++c) {
        const long chunk = blk * CH + c;
        if (chunk >= nchunks) return;
        const long b = chunk * 256 + threadIdx.x;
        if (b >= B) return;
        double v[NIN];
#pragma unroll
        for (int k = 0; k < NIN; ++k) v[k] = in[k * B + b];
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < NIN; ++k) s += v[k];
#pragma unroll
        for (int k = 0; k < NOUT; ++k) out[k * B + b] = v[k % NIN] + s * 1e-300;
    }
}

// two consecutive points per lane: 16-byte loads and stores (global_load_dwordx4)
template <int NIN, int NOUT>
__global__ __launch_bounds__(256) void k_stream2(const double* __restrict__ in, double* __restrict__ out, long B) {
    const long b = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
    if (b + 1 >= B) return;
    double2 v[NIN];
#pragma unroll
    for (int k = 0; k < NIN; ++k) v[k] = *(const double2*)(in + k * B + b);
    double2 s = {0.0, 0.0};
#pragma unroll
    for (int k = 0; k < NIN; ++k) { s.x += v[k].x; s.y += v[k].y; }
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
        double2 o = {v[k % NIN].x + s.x * 1e-300, v[k % NIN].y + s.y * 1e-300};
        *(double2*)(out + k * B + b) = o;
    }
}

// non-temporal stores (the outputs are not read again by this kernel)
template <int NIN, int NOUT>
__global__ __launch_bounds__(256) void k_stream_nt(const double* __restrict__ in, double* __restrict__ out, long B) {
    const long b = (long)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    double v[NIN];
#pragma unroll
    for (int k = 0; k < NIN; ++k) v[k] = __builtin_nontemporal_load(in + k * B + b);
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NIN; ++k) s += v[k];
#pragma unroll
    for (int k = 0; k < NOUT; ++k) __builtin_nontemporal_store(v[k % NIN] + s * 1e-300, out + k * B + b);
}

template <int NIN, int NOUT, int MODE>
void run2(const double* in, double* out, long B, const char* name) {
    const long nblocks = MODE == 0 ? (B / 2 + 255) / 256 : (B + 255) / 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&] {
        if (MODE == 0) hipLaunchKernelGGL((k_stream2<NIN, NOUT>), dim3(nblocks), dim3(256), 0, 0, in, out, B);
        else hipLaunchKernelGGL((k_stream_nt<NIN, NOUT>), dim3(nblocks), dim3(256), 0, 0, in, out, B);
    };
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double bytes = 8.0 * (NIN + NOUT) * B;
    printf("%-44s in %2d out %2d %s : %.4f ms  %.2f TB/s\n", name, NIN, NOUT, MODE == 0 ? "16-byte accesses, 2 points/lane" : "non-temporal loads/stores", ms, bytes / ms / 1e9);
}

template <int NIN, int NOUT, int CH, bool XCD>
void run(const double* in, double* out, long B, const char* name) {
    const long nchunks = (B + 255) / 256, nblocks = (nchunks + CH - 1) / CH;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_stream<NIN, NOUT, CH, XCD>), dim3(nblocks), dim3(256), 0, 0, in, out, B, nchunks);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_stream<NIN, NOUT, CH, XCD>), dim3(nblocks), dim3(256), 0, 0, in, out, B, nchunks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double bytes = 8.0 * (NIN + NOUT) * B;
    printf("%-44s in %2d out %2d chunks/block %d xcd %d : %.4f ms  %.2f TB/s\n", name, NIN, NOUT, CH, (int)XCD, ms, bytes / ms / 1e9);
}

int main() {
    const long B = 10000000;
    double *in, *out;
    hipMalloc(&in, 8ul * 22 * B); hipMalloc(&out, 8ul * 13 * B);
    hipMemset(in, 0, 8ul * 22 * B); hipMemset(out, 0, 8ul * 13 * B);
    run<22, 13, 1, false>(in, out, B, "update+vjp shape (176 B in, 104 B out)");
    run<22, 13, 2, false>(in, out, B, "update+vjp shape");
    run<22, 13, 4, false>(in, out, B, "update+vjp shape");
    run<22, 13, 8, false>(in, out, B, "update+vjp shape");
    run<22, 13, 1, true>(in, out, B, "update+vjp shape");
    run<22, 13, 4, true>(in, out, B, "update+vjp shape");
    run<16, 13, 1, false>(in, out, B, "update shape (128 B in, 104 B out)");
    run<16, 13, 4, false>(in, out, B, "update shape");
    run2<22, 13, 0>(in, out, B, "update+vjp shape");
    run2<22, 13, 1>(in, out, B, "update+vjp shape");
    run2<16, 13, 0>(in, out, B, "update shape");
    run2<16, 13, 1>(in, out, B, "update shape");
    run2<8, 8, 0>(in, out, B, "8 rows in, 8 rows out");
    run<22, 1, 1, false>(in, out, B, "objective shape (176 B in, 8 B out)");
    run2<22, 1, 0>(in, out, B, "objective shape");
    run<8, 8, 1, false>(in, out, B, "8 rows in, 8 rows out");
    hipFree(in); hipFree(out);
    return 0;
}
