// What ds_read_b64_tr_b16 (gfx950) hands to each lane for ARBITRARY per-lane addresses: the model the split-bf16 strip
// kernel's second contraction relies on (csrc/fmap.hip).  Each lane of a 16-lane group points at an 8-byte chunk (4 x 16 bit);
// hypothesis: lane l receives, for j < 4, element ((l & 15) & 3) of the chunk that lane 4 j + ((l & 15) >> 2) OF ITS GROUP
// pointed at.  Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/tr16_probe.hip -o /tmp/tr16 && /tmp/tr16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const int* chunk_of_lane, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + chunk_of_lane[l] * 4));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}
int main() {
  int h_chunk[64];
  short h_out[256];
  int *d_chunk; short* d_out;
  hipMalloc(&d_chunk, sizeof(h_chunk)); hipMalloc(&d_out, sizeof(h_out));
  int bad_total = 0;
  for (int trial = 0; trial < 3; ++trial) {
    for (int l = 0; l < 64; ++l)
      h_chunk[l] = trial == 0 ? l                               // the canonical image: chunk l
                 : trial == 1 ? (l >> 4) * 16 + (l & 15) * 37 % 1000   // scattered, distinct
                 : rand() % 1024;                                // random (8-byte aligned by construction)
    hipMemcpy(d_chunk, h_chunk, sizeof(h_chunk), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d_chunk, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const int src_lane = (l & ~15) + 4 * j + ((l & 15) >> 2);
        const int want = h_chunk[src_lane] * 4 + ((l & 15) & 3);
        if (h_out[l * 4 + j] != (short)want) { if (bad < 4) printf("trial %d lane %d j %d: got %d want %d\n", trial, l, j, h_out[l * 4 + j], want); ++bad; }
      }
    printf("trial %d: %d mismatches\n", trial, bad);
    bad_total += bad;
  }
  if (bad_total) { printf("lane 0..3 raw of last trial:"); for (int i = 0; i < 16; ++i) printf(" %d", h_out[i]); printf("\n"); }
  return bad_total != 0;
}
