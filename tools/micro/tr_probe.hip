// tr_probe.hip - which element does lane L of ds_read_b64_tr_b16 receive?  LDS is filled with u16 = its own element index; lane l reads at
// byte address 8 * perm(l) for a few address patterns; prints what each lane got.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void probe(const int* addr, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int a = addr[threadIdx.x];
  bf4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf4*)((__attribute__((address_space(3))) char*)lds + a));
  s4 r = __builtin_bit_cast(s4, v);
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)r[j];
}
int main() {
  int h[64]; unsigned short o[256];
  int* d; unsigned short* od;
  hipMalloc(&d, sizeof(h)); hipMalloc(&od, sizeof(o));
  for (int pat = 0; pat < 3; ++pat) {
    for (int l = 0; l < 64; ++l) {
      if (pat == 0) h[l] = 8 * l;                                   // lane-linear
      if (pat == 1) h[l] = (l & 15) * 144 + (l >> 4) * 32;          // row = lane&15, pitch 144 B
      if (pat == 2) h[l] = ((l & 15) >> 2) * 144 + (l & 3) * 8 + (l >> 4) * 4 * 144;   // 4 rows x 4 chunks per 16 lanes
    }
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, od);
    hipMemcpy(o, od, sizeof(o), hipMemcpyDeviceToHost);
    printf("pattern %d (u16 element indices; address/2 of lane = first column)\n", pat);
    for (int l = 0; l < 64; ++l) printf("lane %2d addr/2 %4d : %4d %4d %4d %4d\n", l, h[l] / 2, o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3]);
  }
  return 0;
}
