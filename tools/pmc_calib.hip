// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the CFAR kernel's access pattern
// (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern
// before trusting an absolute").  One lane moves one dword per row and marches down the rows of
// a rows x cols uint8 frame exactly like cfar_u8_ring (256 contiguous bytes per wave per row,
// row stride = cols bytes); every input byte is read once and every output byte written once,
// so FETCH = WRITE = frames * rows * cols bytes by construction.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pmc_calib tools/pmc_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- /tmp/pmc_calib     (then WRITE_SIZE in its own pass)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(64) void calib_copy_rows(const unsigned *__restrict__ in, unsigned *__restrict__ out,
                                                     int rows, int words_per_row, int rows_per_block)
{
    const int frame = blockIdx.z, chunk = blockIdx.x, tile = blockIdx.y;
    const size_t base = (size_t)frame * rows * words_per_row + (size_t)chunk * 64 + threadIdx.x;
    const int r0 = tile * rows_per_block;
    for (int r = r0; r < r0 + rows_per_block && r < rows; ++r) {
        const size_t i = base + (size_t)r * words_per_row;
        out[i] = in[i] + 1u;
    }
}

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 1024, rows = 1024, cols = 512, wpr = cols / 4, rpb = 52;
    const size_t n = (size_t)frames * rows * cols;
    unsigned *d_in, *d_out;
    if (hipMalloc(&d_in, n) != hipSuccess || hipMalloc(&d_out, n) != hipSuccess)
        return 1;
    hipMemset(d_in, 1, n);
    hipMemset(d_out, 0, n);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const dim3 grid(wpr / 64, (rows + rpb - 1) / rpb, frames);
    for (int it = 0; it < 12; ++it) {
        if (it == 2)
            hipEventRecord(e0);
        hipLaunchKernelGGL(calib_copy_rows, grid, dim3(64), 0, 0, d_in, d_out, rows, wpr, rpb);
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("calib_copy_rows: %zu bytes read + %zu bytes written per launch, %.4f ms/launch, %.0f GB/s\n", n, n, ms / 10,
           2.0 * n / (ms / 10) / 1e6);
    return 0;
}
