"""Recipe: build the REFERENCE's own PointNet++ kernels for gfx950 -> oracle/_ref/libpointnet2_ref_{ieee,fma}.so
(TEST INFRASTRUCTURE ONLY -- the strongest pin the C restatement and the HIP ops of rows 12-17 can get here).

The reference ships the nine ops as CUDA sources (/root/reference/pycontrast/networks/pointnet2/src/*_gpu.cu,
*_gpu.h, cuda_utils.h) behind a `torch.utils.cpp_extension.CUDAExtension` (its setup.py:1-24, `nvcc -O2`).  Its
pybind/THC wrapper files (`*.cpp`, `#include <THC/THC.h>`) no longer build against any current PyTorch, but the
kernel files only need a HIP runtime: on a ROCm PyTorch, `CUDAExtension` translates such sources with the `hipify`
tool that ships inside torch (`torch.utils.hipify`) and hands them to hipcc.  This script does exactly that, by
hand instead of through the reference's setup.py:

  1. `torch.utils.hipify.hipify_python.hipify` reads the sources WHERE THEY LIE and writes its translation into a
     temporary directory outside the repository (deleted afterwards -- no reference source, translated or not,
     enters the tree);
  2. `hipcc --offload-arch=gfx950 -O2` (the reference's optimisation level) compiles the four translated kernel
     files into a shared library, TWICE, because the source form `a*a + b*b + c*c` of the distance kernels
     leaves its rounding to the compiler:
       * `_ieee`: `-ffp-contract=off` -- three rounded products, two rounded sums (what the source says; what an
         `--fmad=false` build computes).  Counterpart of `HCM_CONTRACT_IEEE`.
       * `_fma`:  contraction on (hipcc's and nvcc's default) with `-fno-slp-vectorize`: LLVM's scalar
         contraction `fma(c, c, fma(a, a, b*b))`, which is what a target WITHOUT packed fp32 arithmetic (NVPTX;
         NVVM is the same LLVM combiner) gets from this source.  Counterpart of `HCM_CONTRACT_FMA`.
     (Plain `-O2` on gfx950 is a third rounding: the SLP vectoriser turns two of the three products into one
     `v_pk_mul_f32` and only one multiply-add is fused -- an artefact of CDNA's packed math that no NVIDIA
     build of the reference can show; not built.)
  3. the libraries and a json of their launcher symbols (C++-mangled: the reference declares them without
     `extern "C"`) go to oracle/_ref/ -- git-ignored, not gpurun-ignored, so they travel to the GPU box like the
     product's own `.so` files.

Nothing is stubbed: no header, tool or source file is written by this script.  `oracle/pointnet2_ref.py` binds the
nine `*_kernel_launcher*` entry points (raw pointers + sizes + stream); tests/test_pointnet2_ref_gpu.py runs them
on the MI355X next to the HIP ops and the C restatement.  Without /root/reference (the GPU box) this is a no-op
and the prebuilt libraries are used.
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = '/root/reference/pycontrast/networks/pointnet2/src'
OUT_DIR = os.path.join(HERE, '_ref')
VARIANTS = {'ieee': ['-ffp-contract=off'], 'fma': ['-ffp-contract=fast', '-fno-slp-vectorize']}
OUT_SOS = {k: os.path.join(OUT_DIR, 'libpointnet2_ref_%s.so' % k) for k in VARIANTS}
OUT_SYMS = os.path.join(OUT_DIR, 'pointnet2_ref_symbols.json')
KERNEL_FILES = ['ball_query_gpu', 'group_points_gpu', 'interpolate_gpu', 'sampling_gpu']
LAUNCHERS = ['ball_query_kernel_launcher_fast', 'group_points_kernel_launcher_fast',
             'group_points_grad_kernel_launcher_fast', 'gather_points_kernel_launcher_fast',
             'gather_points_grad_kernel_launcher_fast', 'furthest_point_sampling_kernel_launcher',
             'three_nn_kernel_launcher_fast', 'three_interpolate_kernel_launcher_fast',
             'three_interpolate_grad_kernel_launcher_fast']


def up_to_date():
    outs = list(OUT_SOS.values()) + [OUT_SYMS]
    if not all(os.path.exists(o) for o in outs):
        return False
    if not os.path.isdir(REF_SRC):
        return True
    newest = max(os.path.getmtime(os.path.join(REF_SRC, f)) for f in os.listdir(REF_SRC))
    return min(os.path.getmtime(o) for o in outs) >= max(newest, os.path.getmtime(os.path.abspath(__file__)))


def build(force=False, verbose=False):
    """Returns {variant: path}, or None when neither the reference nor prebuilt libraries are present."""
    if not os.path.isdir(REF_SRC):
        return dict(OUT_SOS) if all(os.path.exists(o) for o in OUT_SOS.values()) else None
    if not force and up_to_date():
        return dict(OUT_SOS)
    from torch.utils import cpp_extension
    from torch.utils.hipify import hipify_python
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp_root = tempfile.mkdtemp(prefix='pn2ref_')
    tmp = os.path.join(tmp_root, 'hipified')           # hipify wants to create its output directory itself
    try:
        devnull = open(os.devnull, 'w')
        old = sys.stdout
        sys.stdout = devnull if not verbose else old            # hipify prints a line per file
        try:
            hipify_python.hipify(project_directory=REF_SRC, output_directory=tmp,
                                 includes=[os.path.join(REF_SRC, '*')], extensions=('.cu', '.h'),
                                 show_detailed=False, is_pytorch_extension=True, hip_clang_launch=True)
        finally:
            sys.stdout = old
            devnull.close()
        srcs = [os.path.join(tmp, f + '.hip') for f in KERNEL_FILES]
        missing = [s for s in srcs if not os.path.exists(s)]
        if missing:
            raise RuntimeError('hipify produced no %s' % missing)
        base = (['hipcc', '--offload-arch=gfx950', '-O2', '-fPIC', '-shared', '-std=c++17',
                 '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1', '-DHIPBLAS_V2']
                + ['-I' + p for p in cpp_extension.include_paths()])
        procs = [(k, subprocess.Popen(base + flags + srcs + ['-o', OUT_SOS[k]], cwd=tmp, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True)) for k, flags in VARIANTS.items()]
        for k, pr in procs:
            out, _ = pr.communicate()
            if pr.returncode != 0:
                raise RuntimeError('reference PointNet++ kernels (%s) did not compile:\n%s' % (k, out))
    finally:
        shutil.rmtree(tmp_root, ignore_errors=True)
    tables = []
    for k, so in OUT_SOS.items():
        nm = subprocess.run(['nm', '-D', '--defined-only', so], capture_output=True, text=True, check=True).stdout
        table = {}
        for line in nm.splitlines():
            sym = line.split()[-1]
            for name in LAUNCHERS:
                if sym.startswith('_Z%d%s' % (len(name), name)):
                    table[name] = sym
        if sorted(table) != sorted(LAUNCHERS):
            raise RuntimeError('launchers missing from the reference build: %s' % sorted(set(LAUNCHERS) - set(table)))
        tables.append(table)
    assert tables[0] == tables[1]
    with open(OUT_SYMS, 'w') as f:
        json.dump(tables[0], f, indent=1, sort_keys=True)
    return dict(OUT_SOS)


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
