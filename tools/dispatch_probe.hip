// How long does the GPU take to DISPATCH workgroups?  (round 5: SQ counters of cf_downsample_radix_kernel showed its
// waves resident ~20 % of the kernel's duration.)  Empty and short kernels at several workgroup shapes, HIP-event time
// per launch.   hipcc --offload-arch=gfx950 -O3 tools/dispatch_probe.hip -o /tmp/dispatch_probe && /tmp/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_empty(int *out)
{
    extern __shared__ int s[];
    if (out && threadIdx.x == 0 && blockIdx.x == 0x7fffffff)
        out[0] = s[0];
}

// every thread spins for `cycles` clock ticks (s_memtime based)
__global__ void k_spin(long long cycles, int *out)
{
    extern __shared__ int s[];
    const long long t0 = clock64();
    while (clock64() - t0 < cycles)
        ;
    if (out && threadIdx.x == 0 && blockIdx.x == 0x7fffffff)
        out[0] = s[0];
}

// persistent form: gridDim.x workgroups loop over n_items, spinning `cycles` per item
__global__ void k_spin_persistent(long long cycles, int n_items, int *out)
{
    extern __shared__ int s[];
    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
        const long long t0 = clock64();
        while (clock64() - t0 < cycles)
            ;
        __syncthreads();
    }
    if (out && threadIdx.x == 0 && blockIdx.x == 0x7fffffff)
        out[0] = s[0];
}

template <class F>
static float timed(F launch, int reps = 20)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i)
        launch();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return 1e3f * ms / reps;
}

int main()
{
    hipFuncSetAttribute((const void *)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    hipFuncSetAttribute((const void *)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    hipFuncSetAttribute((const void *)k_spin_persistent, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    int wall = 0;
    hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0);
    printf("wall clock rate %d kHz (clock64 ticks)\n", wall);
    const int shapes[][3] = {{512, 1024, 0},     {512, 1024, 65536}, {512, 1024, 131072}, {256, 1024, 131072}, {4096, 1024, 0},
                             {4096, 1024, 65536}, {2048, 256, 0},     {2048, 256, 32768},   {8192, 256, 0},      {8192, 64, 0},
                             {512, 512, 131072},  {512, 256, 131072}};
    for (auto &sh : shapes) {
        const float us = timed([&] { hipLaunchKernelGGL(k_empty, dim3(sh[0]), dim3(sh[1]), sh[2], 0, (int *)nullptr); });
        printf("empty  grid %5d x %4d threads, %6d B LDS: %8.2f us per launch = %6.3f us per workgroup, %6.1f ns per wave\n", sh[0],
               sh[1], sh[2], us, us / sh[0], 1e3 * us / (sh[0] * (sh[1] / 64.0)));
    }
    // 20 us of work per workgroup (clock64 = 100 MHz wall clock on gfx9: 2000 ticks)
    const long long ticks = wall > 0 ? (long long)(20e-6 * wall * 1e3) : 2000;
    for (auto &sh : shapes) {
        if (sh[1] != 1024 || sh[0] > 512)
            continue;
        const float us = timed([&] { hipLaunchKernelGGL(k_spin, dim3(sh[0]), dim3(sh[1]), sh[2], 0, ticks, (int *)nullptr); }, 10);
        printf("spin20 grid %5d x %4d threads, %6d B LDS: %8.2f us per launch\n", sh[0], sh[1], sh[2], us);
    }
    for (int g : {256, 512}) {
        const float us = timed([&] { hipLaunchKernelGGL(k_spin_persistent, dim3(g), dim3(1024), 131072, 0, ticks, 512, (int *)nullptr); }, 10);
        printf("spin20 persistent: %d workgroups x 1024 threads over 512 items, 128 KB LDS: %8.2f us per launch\n", g, us);
    }
    return 0;
}
