// Entry points of the fp16-split field path (field_h3.hip), dispatched from the C-ABI in field.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "../../include/nsff_render.h"

int nsff_h3_packed_bytes(const NsffModelDesc* desc, size_t* bytes);
int nsff_h3_pack_weights(const NsffModelDesc* desc, const float* const* params, void* packed, hipStream_t st);
// args already validated by nsff_field_query; points_per_block is 64 or 128
int nsff_h3_field_query(const NsffModelDesc* desc, const void* packed, const NsffFieldArgs* args,
                        int points_per_block, hipStream_t st);

// register-resident f16x3 kernel (field_ra.hip)
int nsff_ra_packed_bytes(const NsffModelDesc* desc, size_t* bytes);
int nsff_ra_pack_weights(const NsffModelDesc* desc, const float* const* params, void* packed, hipStream_t st);
int nsff_ra_field_query(const NsffModelDesc* desc, const void* packed, const NsffFieldArgs* args, hipStream_t st);
