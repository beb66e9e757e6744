// What can the MI355X deliver for the ACCESS PATTERN of the bank pass -- random 512-byte rows out of three [n, 128]
// fp32 matrices -- when nothing else is in the way?  ("settle the bank-pass roofline honestly".)
// Three stripped kernels, each at several table sizes (MALL-resident 3 x 64 MB ... HBM-resident 3 x 2 GB):
//   reg<DEPTH>   16 lanes per row, two float4 per lane and bank (the pass kernel's own load shape), DEPTH row triples
//                in flight per lane group, one add per loaded vector: the register-staged ceiling
//   glds<DEPTH>  the same rows fetched by global_load_lds_dwordx4 straight into LDS (wave-uniform base + lane * 16:
//                one instruction = two rows of one bank), DEPTH stages of 2 row triples per wave in flight, LDS never
//                read back except one word per stage: the DMA-staged ceiling
//   stream       a plain coalesced read of the same number of bytes: the streaming ceiling on this box
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_ceiling gather_ceiling.hip ; run: ./gather_ceiling [out.json]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int D = 128;

template <int DEPTH>
__global__ __launch_bounds__(256) void reg_kernel(const float* __restrict__ b1, const float* __restrict__ b2,
                                                  const float* __restrict__ b3, const int* __restrict__ idx,
                                                  int rows_per_group, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, t = lane & 15;
  const int group = (blockIdx.x * 256 + threadIdx.x) >> 4;
  const int* my = idx + (int64_t)group * rows_per_group;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 ring[DEPTH][6];
  auto load = [&](int j, int k) {
    const int64_t off = (int64_t)my[k] * D + 4 * t;
    ring[j][0] = *reinterpret_cast<const float4*>(b1 + off);
    ring[j][1] = *reinterpret_cast<const float4*>(b1 + off + 64);
    ring[j][2] = *reinterpret_cast<const float4*>(b2 + off);
    ring[j][3] = *reinterpret_cast<const float4*>(b2 + off + 64);
    ring[j][4] = *reinterpret_cast<const float4*>(b3 + off);
    ring[j][5] = *reinterpret_cast<const float4*>(b3 + off + 64);
  };
#pragma unroll
  for (int j = 0; j < DEPTH; ++j) load(j, j);
  for (int k0 = 0; k0 < rows_per_group; k0 += DEPTH) {
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
#pragma unroll
      for (int v = 0; v < 6; ++v) {
        acc.x += ring[j][v].x; acc.y += ring[j][v].y; acc.z += ring[j][v].z; acc.w += ring[j][v].w;
      }
      if (k0 + j + DEPTH < rows_per_group) load(j, k0 + j + DEPTH);
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[group] = acc.x;
}

// r04: the bf16 banks' access pattern -- random 256-BYTE rows (128 bf16), one 16-byte load per lane and
// bank, 16 lanes per row -- with nothing else in the way: what the part gives a gather of half-size rows.
template <int DEPTH>
__global__ __launch_bounds__(256) void reg256_kernel(const uint4* __restrict__ b1, const uint4* __restrict__ b2,
                                                     const uint4* __restrict__ b3, const int* __restrict__ idx,
                                                     int rows_per_group, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, t = lane & 15;
  const int group = (blockIdx.x * 256 + threadIdx.x) >> 4;
  const int* my = idx + (int64_t)group * rows_per_group;
  uint4 acc = make_uint4(0, 0, 0, 0);
  uint4 ring[DEPTH][3];
  auto load = [&](int j, int k) {
    const int64_t off = (int64_t)my[k] * 16 + t;              // a row = 16 uint4
    ring[j][0] = b1[off];
    ring[j][1] = b2[off];
    ring[j][2] = b3[off];
  };
#pragma unroll
  for (int j = 0; j < DEPTH; ++j) load(j, j);
  for (int k0 = 0; k0 < rows_per_group; k0 += DEPTH) {
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
#pragma unroll
      for (int v = 0; v < 3; ++v) { acc.x ^= ring[j][v].x; acc.y ^= ring[j][v].y; acc.z ^= ring[j][v].z; acc.w ^= ring[j][v].w; }
      if (k0 + j + DEPTH < rows_per_group) load(j, k0 + j + DEPTH);
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[group] = (float)acc.x;
}

// one stage = 2 row triples per wave = 6 glds instructions = 3 KB of LDS per wave
template <int DEPTH>
__global__ __launch_bounds__(256) void glds_kernel(const float* __restrict__ b1, const float* __restrict__ b2,
                                                   const float* __restrict__ b3, const int* __restrict__ idx,
                                                   int pairs_per_wave, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gwave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);   // uniform: the row indices come by s_load
  const int* my = idx + (int64_t)gwave * pairs_per_wave * 2;                 // (an ordinary global_load would drain the DMA queue)
  float* base = lds + (size_t)wave * DEPTH * 768;            // 768 floats = 3 KB per stage
  const int half = lane >> 5, col = (lane & 31) * 4;
  float acc = 0.f;
  auto issue = [&](int j, int k) {
    const int r0 = my[2 * k], r1 = my[2 * k + 1];
    const int64_t off = (int64_t)(half ? r1 : r0) * D + col;
    float* dst = base + j * 768;
    __builtin_amdgcn_global_load_lds(b1 + off, (__attribute__((address_space(3))) void*)(dst), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(b2 + off, (__attribute__((address_space(3))) void*)(dst + 256), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(b3 + off, (__attribute__((address_space(3))) void*)(dst + 512), 16, 0, 0);
  };
#pragma unroll
  for (int j = 0; j < DEPTH; ++j) issue(j, j);
  for (int k0 = 0; k0 < pairs_per_wave; k0 += DEPTH) {
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      // oldest stage has landed when at most 3 * (DEPTH - 1) DMA instructions are outstanding
      __builtin_amdgcn_s_waitcnt(0x0F70 | ((3 * (DEPTH - 1)) & 0xF) | ((((3 * (DEPTH - 1)) >> 4) & 0x3) << 14));
      acc += base[j * 768 + lane];
      if (k0 + j + DEPTH < pairs_per_wave) issue(j, k0 + j + DEPTH);
    }
  }
  if (acc == 12345.678f) out[gwave] = acc;
}

__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ p, int64_t n4, float* __restrict__ out) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = p[i];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
}

template <class F>
float time_ms(F&& launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char** argv) {
  FILE* fo = argc > 1 ? fopen(argv[1], "w") : nullptr;
  if (fo) fprintf(fo, "[\n");
  bool first = true;
  auto emit = [&](const char* kern, int depth, int64_t n, int wgs, double gb, double ms) {
    printf("%-8s depth %2d  n %8lld (%5.0f MB x3)  wgs %6d  %7.1f us  %7.1f GB/s\n", kern, depth, (long long)n,
           n * 512.0 / 1e6, wgs, ms * 1e3, gb / (ms * 1e-3));
    if (fo) {
      fprintf(fo, "%s{\"kernel\": \"%s\", \"depth\": %d, \"n_rows\": %lld, \"table_MB_each\": %.0f, \"workgroups\": %d, "
                  "\"us\": %.1f, \"GBps\": %.1f}", first ? "" : ",\n", kern, depth, (long long)n, n * 512.0 / 1e6, wgs,
              ms * 1e3, gb / (ms * 1e-3));
      first = false;
    }
    fflush(stdout);
  };
  const int64_t sizes[] = {131072, 1048576, 4194304};
  const int64_t total_rows = 32LL * 16385 * 2;       // two launches' worth of the headline pass: 1.6 GB gathered
  float* out;
  CK(hipMalloc(&out, 1 << 22));
  for (int64_t n : sizes) {
    float *b1, *b2, *b3;
    CK(hipMalloc(&b1, n * D * 4)); CK(hipMalloc(&b2, n * D * 4)); CK(hipMalloc(&b3, n * D * 4));
    CK(hipMemset(b1, 0, n * D * 4)); CK(hipMemset(b2, 0, n * D * 4)); CK(hipMemset(b3, 0, n * D * 4));
    std::vector<int> h(total_rows);
    uint64_t s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (int)(s % (uint64_t)n); }
    int* idx;
    CK(hipMalloc(&idx, total_rows * 4));
    CK(hipMemcpy(idx, h.data(), total_rows * 4, hipMemcpyHostToDevice));
    // register-staged: groups of 16 lanes; rows_per_group so that the grid is ~8 workgroups per CU
    for (int rows_per_group : {64, 256}) {
      const int groups = (int)(total_rows / rows_per_group), wgs = groups / 16;
      const double gb = (double)wgs * 16 * rows_per_group * 3.0 * 512.0 / 1e9;      // rows actually gathered
      emit("reg", 1, n, wgs, gb, time_ms([&] { reg_kernel<1><<<wgs, 256>>>(b1, b2, b3, idx, rows_per_group, out); }, 10));
      emit("reg", 2, n, wgs, gb, time_ms([&] { reg_kernel<2><<<wgs, 256>>>(b1, b2, b3, idx, rows_per_group, out); }, 10));
      emit("reg", 3, n, wgs, gb, time_ms([&] { reg_kernel<3><<<wgs, 256>>>(b1, b2, b3, idx, rows_per_group, out); }, 10));
      emit("reg", 4, n, wgs, gb, time_ms([&] { reg_kernel<4><<<wgs, 256>>>(b1, b2, b3, idx, rows_per_group, out); }, 10));
      emit("reg", 6, n, wgs, gb, time_ms([&] { reg_kernel<6><<<wgs, 256>>>(b1, b2, b3, idx, rows_per_group, out); }, 10));
      emit("reg", 8, n, wgs, gb, time_ms([&] { reg_kernel<8><<<wgs, 256>>>(b1, b2, b3, idx, rows_per_group, out); }, 10));
    }
    for (int pairs : {64, 256}) {
      const int waves = (int)(total_rows / 2 / pairs), wgs = waves / 4;
      const double gb = (double)wgs * 4 * pairs * 2 * 3.0 * 512.0 / 1e9;
#define GL(DEPTH)                                                                                              \
  {                                                                                                            \
    const size_t ldsb = (size_t)4 * DEPTH * 768 * 4;                                                            \
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(glds_kernel<DEPTH>),                                   \
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));                              \
    emit("glds", DEPTH, n, wgs, gb,                                                                              \
         time_ms([&] { glds_kernel<DEPTH><<<wgs, 256, ldsb>>>(b1, b2, b3, idx, pairs, out); }, 10));            \
  }
      GL(2) GL(3) GL(4) GL(6) GL(8) GL(12)
#undef GL
    }
    {   // 256-byte rows: the same buffers read as [2 n][256 B] tables (the rows gathered are half as long)
      const int64_t n2 = 2 * n;
      std::vector<int> h2(total_rows);
      uint64_t s2 = 0x9E3779B97F4A7C15ull;
      for (auto& v : h2) { s2 ^= s2 << 13; s2 ^= s2 >> 7; s2 ^= s2 << 17; v = (int)(s2 % (uint64_t)n2); }
      CK(hipMemcpy(idx, h2.data(), total_rows * 4, hipMemcpyHostToDevice));
      for (int rows_per_group : {32, 128, 512}) {          // 2048 / 512 / 128 workgroups
        const int groups = (int)(total_rows / rows_per_group), wgs = groups / 16;
        const double gb = (double)wgs * 16 * rows_per_group * 3.0 * 256.0 / 1e9;
        const uint4 *q1 = reinterpret_cast<const uint4*>(b1), *q2 = reinterpret_cast<const uint4*>(b2), *q3 = reinterpret_cast<const uint4*>(b3);
        emit("reg256", 2, n, wgs, gb, time_ms([&] { reg256_kernel<2><<<wgs, 256>>>(q1, q2, q3, idx, rows_per_group, out); }, 10));
        emit("reg256", 4, n, wgs, gb, time_ms([&] { reg256_kernel<4><<<wgs, 256>>>(q1, q2, q3, idx, rows_per_group, out); }, 10));
        emit("reg256", 6, n, wgs, gb, time_ms([&] { reg256_kernel<6><<<wgs, 256>>>(q1, q2, q3, idx, rows_per_group, out); }, 10));
        emit("reg256", 8, n, wgs, gb, time_ms([&] { reg256_kernel<8><<<wgs, 256>>>(q1, q2, q3, idx, rows_per_group, out); }, 10));
        emit("reg256", 12, n, wgs, gb, time_ms([&] { reg256_kernel<12><<<wgs, 256>>>(q1, q2, q3, idx, rows_per_group, out); }, 10));
        emit("reg256", 16, n, wgs, gb, time_ms([&] { reg256_kernel<16><<<wgs, 256>>>(q1, q2, q3, idx, rows_per_group, out); }, 10));
      }
    }
    emit("stream", 0, n, 4096, 3.0 * n * 512.0 / 1e9,
         time_ms([&] { stream_kernel<<<4096, 256>>>(reinterpret_cast<const float4*>(b1), n * D / 4, out);
                       stream_kernel<<<4096, 256>>>(reinterpret_cast<const float4*>(b2), n * D / 4, out);
                       stream_kernel<<<4096, 256>>>(reinterpret_cast<const float4*>(b3), n * D / 4, out); }, 10));
    CK(hipFree(b1)); CK(hipFree(b2)); CK(hipFree(b3)); CK(hipFree(idx));
  }
  if (fo) { fprintf(fo, "\n]\n"); fclose(fo); }
  return 0;
}
