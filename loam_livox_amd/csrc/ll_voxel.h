// ll_voxel.h -- device buffers of the batched VoxelGrid (ll_voxel_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "ll_voxel_core.h"

namespace ll {

struct VoxelDev {
    int max_clouds, stride;
    int out_stride;              // stride of `out` after the last filter call (= the input stride of that call)
    int block_path;              // 1 (default): every cloud of up to 24 576 points is filtered by one workgroup of its own, whatever the batch size (vox_block_kernel);
                                 // 0 (LL_VOXEL_GENERAL_PATH, tests/test_gpu_voxel.py): every cloud through the multi-kernel pipeline, which larger clouds always take
    float4 *in;                  // [max_clouds][stride]  staging for host inputs
    float4 *out;                 // [n_clouds][out_stride] filtered clouds
    int *n, *n_out, *status;     // [max_clouds]
    int *n_vox, *vox_off;        // [max_clouds]
    unsigned int *mm;            // [max_clouds][8] bounding box (ordered encoding) + finite count
    VoxelParams *prm;            // [max_clouds]
    unsigned long long *keys, *keys2;
    unsigned int *vals, *vals2, *is_head, *rank, *head_pos;
    int *n_vox_total;
    void *tmp;
    size_t tmp_bytes;
};

int voxel_alloc(VoxelDev &v, int max_clouds, int stride, const char **err);
void voxel_free(VoxelDev &v);
int voxel_filter(VoxelDev &v, const float4 *in, const int *n, int in_stride, int n_clouds, const float leaf[3], hipStream_t s,
                 const char **err);

}  // namespace ll
