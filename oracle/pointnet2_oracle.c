/* CPU oracle for the PointNet++ point ops.  TEST INFRASTRUCTURE ONLY -- never linked into,
 * imported by or executed from the product package (hcmoco_amd); only tests/, smoke() and the
 * cpu_baseline leg of bench.py may load it.
 *
 * Plain-C sequential restatement of the nine CUDA kernels of the reference
 * (/root/reference/pycontrast/networks/pointnet2/src/*.cu), one function per kernel, citing the
 * lines it follows.  The reference ships no tests or fixtures for these ops and its extension does not
 * build against a current PyTorch (THC), but its kernel files compile for gfx950 from where they lie
 * (oracle/build_ref_pointnet2.py -> oracle/_ref/libpointnet2_ref_{ieee,fma}.so).  PINNED on the GPU by
 * tests/test_pointnet2_ref_gpu.py: every function below is bit-identical to the reference's own kernel
 * (indices, distances, forwards; duplicate-free backwards), in both arithmetic contracts.  On CPU it is
 * additionally held by hand-checkable known-answer tests (tests/test_pointnet2_oracle.py).
 *
 * Arithmetic contract (oracle_set_contract): the reference source evaluates  a*a + b*b + c*c  and is
 * built with `nvcc -O2` (networks/pointnet2/setup.py:20, --fmad=true by default).  Mode 1 (default)
 * restates LLVM's scalar contraction of that source, fma(c, c, fma(a, a, b*b)) (what the reference's
 * kernels compute when built with contraction on and no packed-fp32 vectorisation, i.e. what NVPTX gets)
 * -- taken as the reference's real arithmetic; mode 0 is the un-fused ((a*a + b*b) + c*c) of an
 * --fmad=false / -ffp-contract=off build.  Compiled with -ffp-contract=off: only the
 * fmaf() calls below fuse.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int g_contract = 1;   /* 0 = IEEE un-fused, 1 = FMA (nvcc -O2 default) */
void oracle_set_contract(int c) { g_contract = c ? 1 : 0; }
int oracle_get_contract(void) { return g_contract; }

/* a0*b0 + a1*b1 + a2*b2 in source order under the selected contract */
static float dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
  if (g_contract) {
    volatile float mid = a1 * b1;
    return fmaf(a2, b2, fmaf(a0, b0, mid));
  }
  volatile float xx = a0 * b0, yy = a1 * b1, zz = a2 * b2;
  volatile float s = xx + yy;
  return s + zz;
}

static float sqdist(float ax, float ay, float az, float bx, float by, float bz) {
  volatile float dx = ax - bx, dy = ay - by, dz = az - bz;
  return dot3(dx, dx, dy, dy, dz, dz);
}

/* cuda_utils.h:10-14 */
static int opt_n_threads(int work) {
  int p = 1;
  while ((p << 1) <= work && (p << 1) <= 1024) p <<= 1;
  return p;
}

/* sampling_gpu.cu:93-209.  Simulates the block: thread t scans k = t, t+bs, ... keeping its first
 * strict maximum; the tree (:86-91) keeps the lower thread on ties. */
void oracle_furthest_point_sampling(int b, int n, int m, const float* dataset, float* temp, int* idxs) {
  if (m <= 0) return;
  const int bs = opt_n_threads(n);
  float* dists = (float*)malloc(sizeof(float) * bs);
  int* dists_i = (int*)malloc(sizeof(int) * bs);
  for (int bi = 0; bi < b; ++bi) {
    const float* d = dataset + (size_t)bi * n * 3;
    float* t = temp + (size_t)bi * n;
    int* out = idxs + (size_t)bi * m;
    int old = 0;
    out[0] = old;
    for (int j = 1; j < m; ++j) {
      const float x1 = d[old * 3], y1 = d[old * 3 + 1], z1 = d[old * 3 + 2];
      for (int tid = 0; tid < bs; ++tid) {
        int besti = 0;
        float best = -1.f;
        for (int k = tid; k < n; k += bs) {
          const float dd = sqdist(d[k * 3], d[k * 3 + 1], d[k * 3 + 2], x1, y1, z1);
          const float d2 = dd < t[k] ? dd : t[k];   /* min(d, temp[k]) */
          t[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
        dists[tid] = best;
        dists_i[tid] = besti;
      }
      for (int s = bs / 2; s >= 1; s >>= 1)          /* the unrolled tree, :140-200 */
        for (int tid = 0; tid < s; ++tid) {
          const float v1 = dists[tid], v2 = dists[tid + s];
          const int i1 = dists_i[tid], i2 = dists_i[tid + s];
          dists[tid] = v1 > v2 ? v1 : v2;
          dists_i[tid] = v2 > v1 ? i2 : i1;
        }
      old = dists_i[0];
      out[j] = old;
    }
  }
  free(dists);
  free(dists_i);
}

/* ball_query_gpu.cu:9-45 (idx pre-zeroed by the caller, pointnet2_utils.py:218) */
void oracle_ball_query(int b, int n, int m, float radius, int nsample, const float* new_xyz,
                       const float* xyz, int* idx) {
  const float radius2 = radius * radius;
  for (int bi = 0; bi < b; ++bi)
    for (int pt = 0; pt < m; ++pt) {
      const float* c = new_xyz + ((size_t)bi * m + pt) * 3;
      const float* cloud = xyz + (size_t)bi * n * 3;
      int* out = idx + ((size_t)bi * m + pt) * nsample;
      int cnt = 0;
      for (int k = 0; k < n; ++k) {
        const float d2 = sqdist(c[0], c[1], c[2], cloud[k * 3], cloud[k * 3 + 1], cloud[k * 3 + 2]);
        if (d2 < radius2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) out[l] = k;
          out[cnt] = k;
          ++cnt;
          if (cnt >= nsample) break;
        }
      }
    }
}

/* group_points_gpu.cu:47-66 */
void oracle_group_points(int b, int c, int n, int npoints, int nsample, const float* points,
                         const int* idx, float* out) {
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci)
      for (int p = 0; p < npoints; ++p)
        for (int s = 0; s < nsample; ++s)
          out[(((size_t)bi * c + ci) * npoints + p) * nsample + s] =
              points[((size_t)bi * c + ci) * n + idx[((size_t)bi * npoints + p) * nsample + s]];
}

/* group_points_gpu.cu:8-25 (sequential accumulation order = index order) */
void oracle_group_points_grad(int b, int c, int n, int npoints, int nsample, const float* grad_out,
                              const int* idx, float* grad_points) {
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci)
      for (int p = 0; p < npoints; ++p)
        for (int s = 0; s < nsample; ++s)
          grad_points[((size_t)bi * c + ci) * n + idx[((size_t)bi * npoints + p) * nsample + s]] +=
              grad_out[(((size_t)bi * c + ci) * npoints + p) * nsample + s];
}

/* sampling_gpu.cu:8-24 / :46-63 */
void oracle_gather_points(int b, int c, int n, int m, const float* points, const int* idx, float* out) {
  oracle_group_points(b, c, n, m, 1, points, idx, out);
}
void oracle_gather_points_grad(int b, int c, int n, int m, const float* grad_out, const int* idx,
                               float* grad_points) {
  oracle_group_points_grad(b, c, n, m, 1, grad_out, idx, grad_points);
}

/* interpolate_gpu.cu:9-52: double trackers at 1e40, strict '<', first wins; dist2 stays squared */
void oracle_three_nn(int b, int n, int m, const float* unknown, const float* known, float* dist2, int* idx) {
  for (int bi = 0; bi < b; ++bi)
    for (int pt = 0; pt < n; ++pt) {
      const float* u = unknown + ((size_t)bi * n + pt) * 3;
      const float* kn = known + (size_t)bi * m * 3;
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        const float d = sqdist(u[0], u[1], u[2], kn[k * 3], kn[k * 3 + 1], kn[k * 3 + 2]);
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d; besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d; besti2 = k;
        } else if (d < best3) {
          best3 = d; besti3 = k;
        }
      }
      float* o = dist2 + ((size_t)bi * n + pt) * 3;
      int* oi = idx + ((size_t)bi * n + pt) * 3;
      o[0] = (float)best1; o[1] = (float)best2; o[2] = (float)best3;
      oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
    }
}

/* interpolate_gpu.cu:77-97 */
void oracle_three_interpolate(int b, int c, int m, int n, const float* points, const int* idx,
                              const float* weight, float* out) {
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci)
      for (int pt = 0; pt < n; ++pt) {
        const float* w = weight + ((size_t)bi * n + pt) * 3;
        const int* id = idx + ((size_t)bi * n + pt) * 3;
        const float* p = points + ((size_t)bi * c + ci) * m;
        out[((size_t)bi * c + ci) * n + pt] = dot3(w[0], p[id[0]], w[1], p[id[1]], w[2], p[id[2]]);
      }
}

/* interpolate_gpu.cu:120-142 */
void oracle_three_interpolate_grad(int b, int c, int n, int m, const float* grad_out, const int* idx,
                                   const float* weight, float* grad_points) {
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci)
      for (int pt = 0; pt < n; ++pt) {
        const float* w = weight + ((size_t)bi * n + pt) * 3;
        const int* id = idx + ((size_t)bi * n + pt) * 3;
        float* g = grad_points + ((size_t)bi * c + ci) * m;
        const float go = grad_out[((size_t)bi * c + ci) * n + pt];
        g[id[0]] += go * w[0];
        g[id[1]] += go * w[1];
        g[id[2]] += go * w[2];
      }
}
