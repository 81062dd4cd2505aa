// Entry points of the fp16-split field path (field_h3.hip), dispatched from the C-ABI in field.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "../../include/nsff_render.h"

int nsff_h3_packed_bytes(const NsffModelDesc* desc, size_t* bytes);
int nsff_h3_pack_weights(const NsffModelDesc* desc, const float* const* params, void* packed, bool fold, hipStream_t st);
int nsff_h3_fold_heads(const NsffModelDesc* desc, const float* const* params, void* packed, hipStream_t st);
int nsff_fold_rows_f32(const NsffModelDesc* desc, const float* const* params, float* s_w, float* s_b, float* t_w, float* t_b,
                       hipStream_t st);
// args already validated by nsff_field_query; points_per_block is 64 / 130 / 131 (the f16x3 tilings of NsffFieldArgs::tile_points)
// span (or null, profiling): NSFF_SPAN_WORDS zeroed uint64 that receive the summed lifetimes of a sample of the launch's
// workgroups in shader-clock ticks [0] and in wall-clock ticks [1]
#define NSFF_SPAN_WORDS 2
int nsff_h3_field_query(const NsffModelDesc* desc, const void* packed, const NsffFieldArgs* args,
                        int points_per_block, hipStream_t st, unsigned long long* span = nullptr);

extern int g_nsff_last_h3_kernel;
extern int g_nsff_last_h3_grid;
