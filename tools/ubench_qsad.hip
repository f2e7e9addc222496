// ubench_qsad.hip -- semantics check of V_QSAD_PK_U16_U8 / V_MQSAD_PK_U16_U8 / V_SAD_U8 / V_MSAD_U8 on gfx950 (the block matcher relies on them).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint64_t *out, uint64_t s0, uint32_t s1, uint64_t s2) {
  out[0] = __builtin_amdgcn_qsad_pk_u16_u8(s0, s1, s2);
  out[1] = __builtin_amdgcn_mqsad_pk_u16_u8(s0, s1, s2);
  out[2] = __builtin_amdgcn_sad_u8((uint32_t)s0, s1, 5);
  out[3] = __builtin_amdgcn_msad_u8((uint32_t)s0, s1, 5);
}
int main() {
  uint64_t *d; hipMalloc(&d, 64);
  // bytes of s0: 10,20,30,40,50,60,70,80 ; s1 bytes: 12, 0, 33, 44
  uint64_t s0 = 0; uint8_t b0[8] = {10,20,30,40,50,60,70,80}; for (int i = 0; i < 8; ++i) s0 |= (uint64_t)b0[i] << (8*i);
  uint32_t s1 = 12 | (0u << 8) | (33u << 16) | (44u << 24);
  uint64_t s2 = 1 | (2ull << 16) | (3ull << 32) | (4ull << 48);
  k<<<1,1>>>(d, s0, s1, s2);
  uint64_t h[4]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  for (int j = 0; j < 2; ++j) { printf("%s:", j ? "mqsad" : "qsad"); for (int i = 0; i < 4; ++i) printf(" %u", (unsigned)((h[j] >> (16*i)) & 0xffff)); printf("\n"); }
  printf("sad %llu msad %llu\n", (unsigned long long)h[2], (unsigned long long)h[3]);
  // expected qsad[i] = sum_j |b0[i+j] - s1[j]| + acc[i]
  int s1b[4] = {12,0,33,44};
  for (int i = 0; i < 4; ++i) { int a = 0, m = 0, m2 = 0; for (int j = 0; j < 4; ++j) { int t = abs((int)b0[i+j]-s1b[j]); a += t; if (s1b[j]) m += t; if (b0[i+j]) m2 += t; } printf("i=%d full %d  masked-by-s1 %d (+acc %d)\n", i, a + (i+1), m + (i+1), i+1); }
  return 0;
}
