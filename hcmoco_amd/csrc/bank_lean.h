// Internal: the instruction-lean form of the fused bank pass (csrc/bank_lean.hip), launched from csrc/bank.hip.
#pragma once
#include <cstdint>

namespace hcm {
// ring: register-ring depth (bf16: 4, 5, 6, 8; fp32: 2, 3, 4).  Returns hipGetLastError() of the launch.
int bank_pass_lean_launch(int is_bf16, int ring, const void* b1, const void* b2, const void* b3, const int64_t* idx,
                          const float* x1, const float* x2, const float* x3, int B, int K1, int R, float scale,
                          float* part_m, float* part_s, float* part_acc, float* l0, void* stream);
}  // namespace hcm
