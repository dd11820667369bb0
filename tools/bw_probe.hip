// Bandwidth probe for the CFAR traversal: a copy kernel that marches down the range rows of a frame batch
// with 1 / 2 / 4 dwords per lane per row (256 B / 512 B / 1 KiB per wave per row).  Measured on MI355X
// (1 GiB per launch): W=1 4.1-4.6 TB/s, W=2 4.3-5.0 TB/s, W=4 5.1 TB/s -- the row-march pattern itself tops out
// near 5 TB/s (a linear float4 stream reaches 6.3), which is where cfar_u8_ring runs (5.0-5.2 TB/s).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bw tools/bw_probe.hip && /tmp/bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
// row-march copy: each lane moves W dwords per row, W*256 B per wave per row, rows of `cols` bytes
template <int W>
__global__ __launch_bounds__(64) void copy_rows(const unsigned *__restrict__ in, unsigned *__restrict__ out, int rows,
                                               int words_per_row, int rows_per_block)
{
    const int frame = blockIdx.z, chunk = blockIdx.x, tile = blockIdx.y;
    const size_t base = (size_t)frame * rows * words_per_row + (size_t)chunk * 64 * W + threadIdx.x * W;
    const int r0 = tile * rows_per_block;
    for (int r = r0; r < r0 + rows_per_block && r < rows; ++r) {
        const size_t i = base + (size_t)r * words_per_row;
        if (W == 1) out[i] = in[i] + 1u;
        if (W == 2) { uint2 v = *(const uint2 *)(in + i); v.x += 1; *(uint2 *)(out + i) = v; }
        if (W == 4) { uint4 v = *(const uint4 *)(in + i); v.x += 1; *(uint4 *)(out + i) = v; }
    }
}
template <int W> void run(unsigned *d_in, unsigned *d_out, int frames, int rows, int cols, int rpb, int threads_note)
{
    const int wpr = cols / 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(wpr / (64 * W), (rows + rpb - 1) / rpb, frames);
    for (int it = 0; it < 12; ++it) {
        if (it == 2) hipEventRecord(e0);
        hipLaunchKernelGGL(copy_rows<W>, grid, dim3(64), 0, 0, d_in, d_out, rows, wpr, rpb);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)frames * rows * cols;
    printf("W=%d dwords/lane cols=%d rows/block=%d: %.4f ms/launch, %.0f GB/s\n", W, cols, rpb, ms / 10, 2.0 * n / (ms / 10) / 1e6);
}
int main()
{
    const int frames = 1024, rows = 1024, cols = 512;
    const size_t n = (size_t)frames * rows * cols;
    unsigned *d_in, *d_out;
    hipMalloc(&d_in, n); hipMalloc(&d_out, n); hipMemset(d_in, 1, n); hipMemset(d_out, 0, n);
    for (int rpb : {52, 104, 256}) {
        run<1>(d_in, d_out, frames, rows, cols, rpb, 0);
        run<2>(d_in, d_out, frames, rows, cols, rpb, 0);
    }
    // 1024-wide frames (config B shape, same bytes): W=1,2,4
    run<1>(d_in, d_out, 256, 2048, 1024, 104, 0);
    run<2>(d_in, d_out, 256, 2048, 1024, 104, 0);
    run<4>(d_in, d_out, 256, 2048, 1024, 104, 0);
    return 0;
}
