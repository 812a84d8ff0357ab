// Probe of ds_read_b64_tr_b16 on gfx950: which element does lane t of a 16-lane group receive when lane t' supplies the
// address of row (t' >> 2), columns 4 (t' & 3) .. +3 of a [4][16] 16-bit block?   hipcc --offload-arch=gfx950 tr_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[4 * 64 * 4];   // 4 blocks of [4 rows][64 cols]; value = row * 64 + col + 1000 * block
  for (int i = threadIdx.x; i < 4 * 64 * 4; i += 64) lds[i] = (short)((i / 256) * 1000 + (i % 256));
  __syncthreads();
  const int lane = threadIdx.x, t = lane & 15, grp = lane >> 4;
  // group g reads block g: row (t >> 2), cols 16 * g + 4 (t & 3)   (row stride 64 elements)
  const short* p = lds + grp * 256 + (t >> 2) * 64 + 16 * grp + 4 * (t & 3);
  typedef __attribute__((address_space(3))) s16x4 lds4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4*)p);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  probe<<<1, 64>>>(d);
  short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) {
    const int t = lane & 15, g = lane >> 4;
    printf("lane %2d:", lane);
    for (int e = 0; e < 4; ++e) {
      const int want = g * 1000 + e * 64 + 16 * g + t;   // expected: row e, column 16 g + t of block g
      printf(" %5d%s", h[lane * 4 + e], h[lane * 4 + e] == want ? "" : "!");
      bad += h[lane * 4 + e] != want;
    }
    printf("\n");
  }
  printf(bad ? "MISMATCH %d\n" : "tr16 semantics as assumed: lane t gets column t of the 4 rows\n", bad);
  return 0;
}
