// cost of a grid-wide barrier on MI355X (cooperative launch, cooperative_groups grid sync): N syncs inside one kernel, timed with events.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/grid_sync_bench tools/grid_sync_bench.hip     run: tools/grid_sync_bench [blocks] [threads]
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
namespace cg = cooperative_groups;
__global__ void k_syncs(int n, double *buf, long len) {
    cg::grid_group g = cg::this_grid();
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long)gridDim.x * blockDim.x;
    for (int s = 0; s < n; s++) {
        for (long q = tid; q < len; q += nt) buf[q] = buf[q] * 0.5 + 1.0;     // a little work that crosses blocks between syncs
        g.sync();
    }
}
__global__ void k_empty(double *buf) { if (buf == nullptr) buf[0] = 0; }
int main(int argc, char **argv) {
    int blocks = argc > 1 ? atoi(argv[1]) : 256, threads = argc > 2 ? atoi(argv[2]) : 256;
    double *buf; long len = 1 << 20;
    hipMalloc(&buf, sizeof(double) * len); hipMemset(buf, 0, sizeof(double) * len);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int n : {1, 101, 1001}) {
        void *args[] = {&n, &buf, &len};
        hipError_t e = hipLaunchCooperativeKernel((const void *)k_syncs, dim3(blocks), dim3(threads), args, 0, 0);
        if (e != hipSuccess) { printf("cooperative launch refused: %s\n", hipGetErrorString(e)); return 1; }
        hipDeviceSynchronize();
        hipEventRecord(a, 0);
        e = hipLaunchCooperativeKernel((const void *)k_syncs, dim3(blocks), dim3(threads), args, 0, 0);
        hipEventRecord(b, 0); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("blocks %d x %d threads, %4d rounds of (8 MB touched + grid sync): %.1f us total, %.2f us per round\n", blocks, threads, n, ms * 1e3, ms * 1e3 / n);
    }
    // back-to-back tiny launches for comparison
    hipEventRecord(a, 0);
    for (int q = 0; q < 1000; q++) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(threads), 0, 0, buf);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("1000 empty launches of the same shape: %.2f us each\n", ms);
    return 0;
}
