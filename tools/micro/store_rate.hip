// Micro-benchmark: cost of a burst of global stores per CU, by address pattern (run on the GPU box).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/store_rate.hip -o gpurun_out/store_rate && gpurun_out/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// pattern: rows per instruction R (each row gets 1024/R contiguous bytes for x4, 512/R for x2), row stride LD bytes
template <int WIDTH>
__global__ __launch_bounds__(512) void k(unsigned char* out, int R, long ld, int ninstr, int waves_active, int iters, unsigned long long* t, long wg_stride) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bytes_per_row = 64 * WIDTH / R, lanes_per_row = 64 / R;
  unsigned char* base = out + (size_t)blockIdx.x * wg_stride + (size_t)wave * (size_t)ninstr * R * ld
                        + (size_t)(lane / lanes_per_row) * ld + (size_t)(lane % lanes_per_row) * WIDTH;
  (void)bytes_per_row;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), t1 = t0, t2 = t0;
  for (int it = 0; it < iters; ++it) {
    if (wave < waves_active) {
      for (int i = 0; i < ninstr; ++i) {
        unsigned char* p = base + (size_t)i * R * ld;
        if (WIDTH == 16) *reinterpret_cast<u32x4*>(p) = u32x4{(unsigned)i, 1u, 2u, (unsigned)it};
        else *reinterpret_cast<u32x2*>(p) = u32x2{(unsigned)i, (unsigned)it};
      }
    }
    if (it == iters - 1) t1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  t2 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { t[blockIdx.x * 2] = t2 - t0; t[blockIdx.x * 2 + 1] = t1 - t0; }
}

int main() {
  const int G = 256;
  size_t bytes = (size_t)1 << 30;
  unsigned char* out; unsigned long long* t;
  hipMalloc(&out, bytes); hipMalloc(&t, G * 16);
  struct Cfg { const char* name; int width, R; long ld; int ninstr, waves; long wg_stride; };
  std::vector<Cfg> cfgs = {
    {"x4  8 rows x128B ld4096, 16/wave, 8 waves (gemm bf16 tile)", 16, 8, 4096, 16, 8, 1 << 20},
    {"x4  8 rows x128B ld4096, 16/wave, 8 waves, all WG same addr", 16, 8, 4096, 16, 8, 0},
    {"x4  1 row x1024B contiguous, 16/wave, 8 waves", 16, 1, 1024, 16, 8, 1 << 20},
    {"x4  2 rows x512B ld4096, 16/wave, 8 waves", 16, 2, 4096, 16, 8, 1 << 20},
    {"x4  4 rows x256B ld8192, 16/wave, 8 waves (gemm f32 tile /2)", 16, 4, 8192, 16, 8, 1 << 20},
    {"x4 16 rows x64B ld4096, 16/wave, 8 waves", 16, 16, 4096, 16, 8, 2 << 20},
    {"x4 64 rows x16B ld4096, 16/wave, 8 waves", 16, 64, 4096, 4, 8, 2 << 20},
    {"x2 16 rows x32B ld4096, 32/wave, 8 waves (direct bf16x4)", 8, 16, 4096, 32, 8, 1 << 21},
    {"x4  8 rows x128B ld4096, 16/wave, 4 waves", 16, 8, 4096, 16, 4, 1 << 20},
    {"x4  8 rows x128B ld4096, 32/wave, 4 waves", 16, 8, 4096, 32, 4, 1 << 20},
    {"x4  8 rows x128B ld4096, 16/wave, 1 wave", 16, 8, 4096, 16, 1, 1 << 20},
    {"x4  8 rows x128B ld4096, 128/wave, 1 wave", 16, 8, 4096, 128, 1, 1 << 20},
  };
  for (auto& c : cfgs) {
    for (int grid : {1, 256}) {
      const int iters = 20;
      for (int rep = 0; rep < 2; ++rep) {
        if (c.width == 16) hipLaunchKernelGGL(k<16>, dim3(grid), dim3(512), 0, 0, out, c.R, c.ld, c.ninstr, c.waves, iters, t, c.wg_stride);
        else hipLaunchKernelGGL(k<8>, dim3(grid), dim3(512), 0, 0, out, c.R, c.ld, c.ninstr, c.waves, iters, t, c.wg_stride);
      }
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(grid * 2);
      hipMemcpy(h.data(), t, grid * 16, hipMemcpyDeviceToHost);
      double tot = 0;
      for (int b = 0; b < grid; ++b) tot += (double)h[b * 2];
      double us_per_burst = tot / grid / iters / 100.0;
      double kb = (double)c.ninstr * c.waves * 64 * c.width / 1024.0;
      printf("%-62s grid %3d: %7.2f us per burst of %5.0f KB (%3d instr/CU)  = %6.1f ns/instr  %6.1f GB/s/CU\n", c.name, grid, us_per_burst, kb,
             c.ninstr * c.waves, us_per_burst * 1000 / (c.ninstr * c.waves), kb * 1024 / us_per_burst / 1000);
    }
  }
  return 0;
}
