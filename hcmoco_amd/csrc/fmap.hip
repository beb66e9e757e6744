// placeholder, replaced by the real kernels
#include "hcm_common.h"
#include "../../include/hcmoco_hip.h"
extern "C" {
size_t hcm_dense_soft_nce_workspace_bytes(int, int, int) { return 0; }
int hcm_dense_soft_nce(const float*, const float*, hcm_strides4, int, int, int, int, const int64_t*, const int32_t*, int, float, float*, float*, float*, void*, size_t, hcm_stream_t) { return (int)hipErrorNotSupported; }
size_t hcm_joint_nce_workspace_bytes(int, int, int) { return 0; }
int hcm_joint_nce(const float*, const float*, hcm_strides4, int, int, int, int, const float*, const int64_t*, const int32_t*, const int32_t*, int, float, float*, float*, float*, float*, void*, size_t, hcm_stream_t) { return (int)hipErrorNotSupported; }
size_t hcm_scl_workspace_bytes(int, int, int) { return 0; }
int hcm_scl(const float*, const float*, hcm_strides4, int, int, int, int, const int64_t*, const int32_t*, const int32_t*, int, float, float*, float*, float*, void*, size_t, hcm_stream_t) { return (int)hipErrorNotSupported; }
int hcm_joint_pixels(const float*, int, int, int64_t*, hcm_stream_t) { return (int)hipErrorNotSupported; }
}
