// Store-path microbenchmark (round 5): what one CU's waves can push into L2 / HBM per clock, by store width and wave count.
// hipcc --offload-arch=gfx950 -O3 -o store_rate store_rate.hip && ./store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// MODE 0: dwordx4 (1 KiB per wave instruction), 1: dwordx4 nontemporal, 2: dwordx2 x 2, 3: dword x 4, 4: dwordx4 but a lane writes 4 consecutive
// rows' 4-byte... (not used).  Every workgroup writes `rows` rows of 1 KiB at pitch `pitch` bytes; wave w takes rows w, w + W, ...
template <int MODE>
__global__ __launch_bounds__(768) void store_kernel(float* out, int rows, int pitch_f, int active_waves, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= active_waves) return;
  f4 v = {1.f * lane, 2.f, 3.f, 4.f};
  for (int rep = 0; rep < reps; ++rep) {
    float* base = out + ((size_t)blockIdx.x * rows) * pitch_f + (size_t)(rep & 1) * 256;
    for (int r = wave; r < rows; r += active_waves) {
      float* p = base + (size_t)r * pitch_f + lane * 4;
      if (MODE == 0) *reinterpret_cast<f4*>(p) = v;
      else if (MODE == 1) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p));
      else if (MODE == 2) { float* q = base + (size_t)r * pitch_f + lane * 2; *reinterpret_cast<f2*>(q) = f2{v[0], v[1]}; *reinterpret_cast<f2*>(q + 128) = f2{v[2], v[3]}; }
      else if (MODE == 3) { float* q = base + (size_t)r * pitch_f + lane; q[0] = v[0]; q[64] = v[1]; q[128] = v[2]; q[192] = v[3]; }
      else {
        // the MFMA accumulator layout stored straight from registers: one instruction = 4 rows x 16 consecutive floats (64 B segments); 16
        // instructions cover a wave's 4 rows x 256 columns (here: rows 4 r .. 4 r + 3 of the workgroup's tile, r = the wave's row group)
        if (4 * r + 3 < rows) {
          float* q = base + (size_t)(4 * r + (lane >> 4)) * pitch_f + (lane & 15);
#pragma unroll
          for (int j = 0; j < 16; ++j) q[16 * j] = v[j & 3];
        }
      }
      v[0] += 1.f;
    }
  }
}

template <int MODE>
void run(const char* name, float* out, int blocks, int rows, int pitch_f, int waves, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  store_kernel<MODE><<<blocks, 768>>>(out, rows, pitch_f, waves, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  store_kernel<MODE><<<blocks, 768>>>(out, rows, pitch_f, waves, reps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)blocks * rows * 1024.0 * reps;
  printf("%-34s blocks %4d waves %2d rows %4d reps %3d: %8.1f us  %7.1f GB/s  %6.2f B/ns/CU\n", name, blocks, waves, rows, reps, ms * 1e3, bytes / ms / 1e6,
         bytes / blocks / (ms * 1e6));
}

int main() {
  float* out;
  const int pitch_f = 512;                                        // 2 KiB row pitch (E = 512 fp32), a workgroup writes the left KiB of its rows
  hipMalloc(&out, (size_t)256 * 4096 * pitch_f * 4 + 4096);
  for (int blocks : {256, 1})
    for (int waves : {12, 8}) {
      run<0>("dwordx4", out, blocks, 160, pitch_f, waves, 64);
      run<1>("dwordx4 nontemporal", out, blocks, 160, pitch_f, waves, 64);
      run<2>("dwordx2 x 2", out, blocks, 160, pitch_f, waves, 64);
      run<3>("dword x 4", out, blocks, 160, pitch_f, waves, 64);
      run<4>("dword, 4 rows x 64 B per instr", out, blocks, 160, pitch_f, waves, 64);
    }
  // a longer stream (rows per workgroup x4: no re-writing of the same lines)
  run<0>("dwordx4, 2560 rows once", out, 256, 2560, pitch_f, 12, 1);
  run<1>("dwordx4 nt, 2560 rows once", out, 256, 2560, pitch_f, 12, 1);
  return 0;
}
