// N2: plane splatting + MPI compositing for time interpolation (reference models/rendering.py:365-460,
// models/softsplat.py:6-44,303-326).  Scatter work without global atomics: output blocks (32 x 8 pixels x 4 planes) are OWNED
// by workgroups.
//   splat_tiles_kernel   accumulates the NEAR samples (landing within a 4-pixel halo of their own pixel) of the block's
//                        surroundings in LDS (ds_add_f64: see lds_add) and writes the block with plain stores; while it projects its own
//                        samples anyway it counts the FAR ones per destination block (LDS histogram -> one global add per
//                        (workgroup, destination));
//   splat_scan_kernel    exclusive scan of the destination counts -> where each block's records start;
//   splat_bin_kernel     workgroups that own far samples re-project them and append one 32-byte record {landing, plane,
//                        rgba} per (sample, destination block) to the destination's range;
//   splat_gather_kernel  one workgroup per destination block with records: LDS accumulate, then add to the block.
// Samples that move farther than the halo are rare for a trained flow field (the last three kernels then return at once);
// random-init flow heads move EVERY sample ~50 px, which the first version paid for with 20 device-scope atomics per sample
// (49 ms per 512 x 288 x 256 frame at ~15 G atomics/s).  Without a workspace (or on frames of more than MAX_TILES blocks)
// the far samples still take that route (splat_far_kernel).
// Compositing is one wavefront per pixel with a segmented product scan over the planes (same scheme as composite_kernel
// in rays.hip).
#include "nsff_common.h"

namespace {

constexpr int WAVES_PER_BLOCK = 4;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float u = __shfl_up(v, off);
        if (lane >= off) v *= u;
    }
    return v;
}

// datasets/ray_utils.py:127-151, (N,3) branch, same operation order
__device__ __forceinline__ void ndc2world(const float x, const float y, const float z, const float* K4, float* w) {
    const float rz = 2.0f / (z - 1.0f - 1e-6f);
    w[0] = -rz * x * K4[2] / K4[0];
    w[1] = -rz * y * K4[3] / K4[1];
    w[2] = rz;
}

// Where sample (pixel px,py; plane s) of the source frame lands: bilinear corner (nwx, nwy) and the four weights.
struct Landing { int nwx, nwy; float w[4]; float ox, oy; bool finite; };

// (xp: the sample's NDC point, fp: its scene flow -- already in registers)
__device__ __forceinline__ Landing project_values(const NsffSplatArgs& a, const float (&xp)[3], const float (&fp)[3], int px, int py) {
    const float x = xp[0], y = xp[1], z = xp[2];
    float pw[3], qw[3];
    ndc2world(x, y, z, a.K4, pw);
    ndc2world(x + fp[0], y + fp[1], z + fp[2], a.K4, qw);
#pragma unroll
    for (int c = 0; c < 3; ++c) qw[c] = pw[c] + a.scale * (qw[c] - pw[c]);
    float uvd[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        uvd[r] = (a.P[4 * r] * qw[0] + a.P[4 * r + 1] * qw[1] + a.P[4 * r + 2] * qw[2]) + a.P[4 * r + 3];
    // optical flow on this plane, then the splat target (rendering.py:411-414, softsplat.py:16-17)
    const float ox = (float)px + (uvd[0] / uvd[2] - (float)px);
    const float oy = (float)py + (uvd[1] / uvd[2] - (float)py);
    const float flx = floorf(ox), fly = floorf(oy);
    Landing L;
    L.ox = ox; L.oy = oy;
    L.finite = flx >= -2.0f && flx <= (float)a.W && fly >= -2.0f && fly <= (float)a.H;   // false for NaN / far outside
    L.nwx = L.finite ? (int)flx : -4;
    L.nwy = L.finite ? (int)fly : -4;
    const float sex = (float)(L.nwx + 1), sey = (float)(L.nwy + 1);
    L.w[0] = (sex - ox) * (sey - oy);                 // north-west
    L.w[1] = (ox - (float)L.nwx) * (sey - oy);        // north-east
    L.w[2] = (sex - ox) * (oy - (float)L.nwy);        // south-west
    L.w[3] = (ox - (float)L.nwx) * (oy - (float)L.nwy);
    return L;
}

__device__ __forceinline__ Landing project_sample(const NsffSplatArgs& a, long long idx, int px, int py) {
    const float* xq = a.xyz + idx * 3;
    const float* fq = a.flow + idx * 3;
    const float xp[3] = {xq[0], xq[1], xq[2]}, fp[3] = {fq[0], fq[1], fq[2]};
    return project_values(a, xp, fp, px, py);
}

// the four bilinear weights of a landing, exactly as project_sample forms them
__device__ __forceinline__ void landing_weights(Landing& L) {
    const float sex = (float)(L.nwx + 1), sey = (float)(L.nwy + 1);
    L.w[0] = (sex - L.ox) * (sey - L.oy);
    L.w[1] = (L.ox - (float)L.nwx) * (sey - L.oy);
    L.w[2] = (sex - L.ox) * (L.oy - (float)L.nwy);
    L.w[3] = (L.ox - (float)L.nwx) * (L.oy - (float)L.nwy);
}

// A sample is "near" when its landing cell is within HALO pixels of its own pixel: then every output tile it
// touches sees it inside its halo and accumulates it in LDS.  Everything else goes through global atomics.
// LDS accumulators are DOUBLES: on gfx950 ds_add_f32 retires ~0.4 lanes per clock per CU while ds_add_f64 does 8.6 and
// ds_add_u32 16 (tools/debug/probes/lds_atomic_rate*.hip) -- the f32 form made this pass atomic-bound at exactly that rate
// (174 G lane-adds/s).  The products are formed in fp32 as the reference's kernel forms them (softsplat.py:27-43); their sum in
// double is the exact-order-independent one, rounded once when the block is written.  Twice the bytes per cell: four planes
// per workgroup instead of eight keep the accumulator at 40 KiB.
__device__ __forceinline__ void lds_add(double* cell, float v) { atomicAdd(cell, (double)v); }
#ifndef SPLAT_PL
#define SPLAT_PL 4
#endif
#ifndef SPLAT_MAX_TILES
#define SPLAT_MAX_TILES 3072
#endif
constexpr int TILE_X = 32, TILE_Y = 8, HALO = 4, PL = SPLAT_PL;     // output tile, halo, planes per workgroup
constexpr int REG_X = TILE_X + 2 * HALO, REG_Y = TILE_Y + 2 * HALO;
__device__ __forceinline__ bool is_near(const Landing& L, int px, int py) {
    const int dx = L.nwx - px, dy = L.nwy - py;
    return L.finite && dx >= -HALO && dx < HALO && dy >= -HALO && dy < HALO;
}

// Pass 1: each workgroup OWNS a TILE_X x TILE_Y block of output pixels for PL consecutive planes, accumulates the
// near samples of the surrounding (tile + halo) source region in LDS (ds_add_f64) and writes the block with plain
// stores -- no global atomics, no memset; the 2.5x redundant projection work is cheap next to the atomics it saves.
// Workspace of the binned far path (ints, then 32-byte records): per block b = tile + n_tiles * plane group
struct FarWork {
    int* count;        // [n_blocks] far records destined for block b
    int* start;        // [n_blocks + 1] exclusive scan of count
    int* cursor;       // [n_blocks] next free record of block b (splat_bin_kernel)
    int* own;          // [n_blocks] far records the samples OWNED by block b produce
    float* records;    // capacity x 8 floats: {ox, oy, plane in group (int bits), r | g, b, a, -} (landing as project_sample left it)
    long long capacity;
    int n_tiles, tiles_x, n_blocks;
};
constexpr int MAX_TILES = SPLAT_MAX_TILES;      // LDS histogram of destination tiles (12 KiB) next to the 40 KiB accumulator: three workgroups per CU

__device__ __forceinline__ FarWork far_work(const NsffSplatArgs& a) {
    FarWork w{};
    w.tiles_x = (a.W + TILE_X - 1) / TILE_X;
    w.n_tiles = w.tiles_x * ((a.H + TILE_Y - 1) / TILE_Y);
    w.n_blocks = w.n_tiles * ((a.n_planes + PL - 1) / PL);
    if (a.work == nullptr || w.n_tiles > MAX_TILES) return w;
    int* p = reinterpret_cast<int*>(a.work);
    const long long ints = 4LL * w.n_blocks + 8;
    w.count = p; w.start = p + w.n_blocks; w.cursor = w.start + w.n_blocks + 1; w.own = w.cursor + w.n_blocks;
    const long long head = (ints * 4 + 31) / 32 * 32;
    w.records = reinterpret_cast<float*>(reinterpret_cast<char*>(a.work) + head);
    w.capacity = (a.work_bytes - head) / 32;
    if (w.capacity < 1) w.count = nullptr;
    return w;
}

// the (up to four) distinct destination tiles of a far landing: calls f(tile) once per tile that holds a corner inside the frame
template <class F>
__device__ __forceinline__ void for_each_dest_tile(const NsffSplatArgs& a, const Landing& L, int tiles_x, F&& f) {
    int seen[4], n = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int cx = L.nwx + (k & 1), cy = L.nwy + (k >> 1);
        if (cx < 0 || cx >= a.W || cy < 0 || cy >= a.H) continue;
        const int t = (cy / TILE_Y) * tiles_x + cx / TILE_X;
        bool dup = false;
        for (int j = 0; j < n; ++j) dup = dup || seen[j] == t;
        if (!dup) { seen[n++] = t; f(t); }
    }
}

// Workgroup -> (output tile, plane group).  The hardware deals workgroups to the eight XCDs round-robin, each XCD has its own
// L2, and what this pass re-reads is spatially local: the 4-pixel halo overlaps the neighbouring tiles, and an 8-plane group
// of one pixel is a 96-byte piece of the 128-byte lines its neighbouring plane groups also touch.  So every XCD gets one
// CONTIGUOUS eighth of the (tile, plane group) sequence, plane groups fastest: neighbours in space are neighbours in
// dispatch order on the same L2.  (SPLAT_XCD=0: the plain 2-D grid order, tiles fastest.)
#ifndef SPLAT_XCD
#define SPLAT_XCD 1
#endif
struct SplatBlock { int tile, pg; bool live; };
__device__ __forceinline__ SplatBlock splat_block(int n_tiles, int n_pg) {
    SplatBlock b;
#if SPLAT_XCD
    const long long total = (long long)n_tiles * n_pg, chunk = (total + 7) / 8;
    const long long k = blockIdx.x >> 3, l = (long long)(blockIdx.x & 7) * chunk + k;
    b.live = k < chunk && l < total;
    b.pg = (int)(l % n_pg); b.tile = (int)(l / n_pg);
#else
    b.live = true; b.tile = blockIdx.x % n_tiles; b.pg = blockIdx.x / n_tiles;
#endif
    return b;
}
__host__ __device__ __forceinline__ int splat_tiles_total(int W, int H) { return ((W + TILE_X - 1) / TILE_X) * ((H + TILE_Y - 1) / TILE_Y); }

__global__ __launch_bounds__(256) void splat_tiles_kernel(const NsffSplatArgs a) {
    __shared__ double sAcc[TILE_X * TILE_Y * PL * 5];          // (double: see lds_add)
    __shared__ int sHist[MAX_TILES];
    const int S = a.n_planes;
    const int tiles_x = (a.W + TILE_X - 1) / TILE_X;
    const SplatBlock blk_ = splat_block(splat_tiles_total(a.W, a.H), (S + PL - 1) / PL);
    if (!blk_.live) return;
    const int tx = blk_.tile % tiles_x, ty = blk_.tile / tiles_x;
    const int x0 = tx * TILE_X, y0 = ty * TILE_Y, s0 = blk_.pg * PL;
    const FarWork fw = far_work(a);
    for (int i = threadIdx.x; i < TILE_X * TILE_Y * PL * 5; i += 256) sAcc[i] = 0.0;
    if (fw.count != nullptr) for (int i = threadIdx.x; i < fw.n_tiles; i += 256) sHist[i] = 0;
    __syncthreads();
    // Four samples per thread and trip: their 16 + 24 bytes each are requested before the first projection -- the pass is a chain
    // of dependent round trips (point + flow, then colour) with three workgroups per CU to hide them, not a bandwidth problem
    // (halving its HBM reads with the XCD order above left its time where it was).
    constexpr int UN = 4, N_ITEMS = REG_X * REG_Y * PL;
    for (int base = threadIdx.x; base < N_ITEMS; base += 256 * UN) {
      float xq[UN][3], fq[UN][3], cq[UN][4];
      long long idxs[UN];
      bool ok[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int item = base + 256 * u;
        const int j = item % PL, rp = item / PL;
        const int px = x0 - HALO + rp % REG_X, py = y0 - HALO + rp / REG_X, s = s0 + j;
        ok[u] = item < N_ITEMS && !(px < 0 || px >= a.W || py < 0 || py >= a.H || s >= S);
        idxs[u] = ok[u] ? ((long long)py * a.W + px) * S + s : 0;
        const float* xg = a.xyz + idxs[u] * 3;
        const float* fg = a.flow + idxs[u] * 3;
        const float* cg = a.rgb + idxs[u] * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) { xq[u][c] = xg[c]; fq[u][c] = fg[c]; cq[u][c] = cg[c]; }
        cq[u][3] = a.alpha[idxs[u]];
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (!ok[u]) continue;
        const int item = base + 256 * u;
        const int j = item % PL, rp = item / PL;
        const int px = x0 - HALO + rp % REG_X, py = y0 - HALO + rp / REG_X;
        const long long idx = idxs[u];
        const Landing L = project_values(a, xq[u], fq[u], px, py);
        if (!is_near(L, px, py)) {
            // a far sample of this block's OWN pixels: one record per destination tile it touches (binned far path)
            const bool own = px >= x0 && px < x0 + TILE_X && py >= y0 && py < y0 + TILE_Y;
            if (own && L.finite && fw.count != nullptr)
                for_each_dest_tile(a, L, tiles_x, [&](int t) { atomicAdd(&sHist[t], 1); });
            continue;
        }
        (void)idx;
        const float src[5] = {cq[u][0], cq[u][1], cq[u][2], cq[u][3], 1.0f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cx = L.nwx + (k & 1) - x0, cy = L.nwy + (k >> 1) - y0;        // inside the owned tile?
            if (cx < 0 || cx >= TILE_X || cy < 0 || cy >= TILE_Y) continue;
            if (L.nwx + (k & 1) >= a.W || L.nwy + (k >> 1) >= a.H) continue;        // (tile may overhang the image)
            double* dst = sAcc + ((cy * TILE_X + cx) * PL + j) * 5;
#pragma unroll
            for (int c = 0; c < 5; ++c) lds_add(dst + c, src[c] * L.w[k]);
        }
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TILE_X * TILE_Y * PL; i += 256) {
        const int j = i % PL, pix = i / PL;
        const int px = x0 + pix % TILE_X, py = y0 + pix / TILE_X, s = s0 + j;
        if (px >= a.W || py >= a.H || s >= S) continue;
        const double* v = sAcc + i * 5;
        float4* dst = reinterpret_cast<float4*>(a.accum + (((long long)py * a.W + px) * S + s) * 8);
        dst[0] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
        dst[1] = make_float4((float)v[4], 0.f, 0.f, 0.f);
    }
    if (fw.count != nullptr) {
        int mine = 0;
        for (int t = threadIdx.x; t < fw.n_tiles; t += 256) {
            const int c = sHist[t];
            if (c) { atomicAdd(fw.count + t + fw.n_tiles * blk_.pg, c); mine += c; }
        }
        if (mine) atomicAdd(fw.own + blk_.tile + fw.n_tiles * blk_.pg, mine);
    }
}

// exclusive scan of the per-block record counts (one workgroup; n_blocks is a few tens of thousands at most)
__global__ __launch_bounds__(1024) void splat_scan_kernel(const NsffSplatArgs a) {
    __shared__ int sWave[16];
    __shared__ int sCarry;
    const FarWork fw = far_work(a);
    if (fw.count == nullptr) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) sCarry = 0;
    __syncthreads();
    for (int base = 0; base < fw.n_blocks; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < fw.n_blocks ? fw.count[i] : 0;
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(incl, off); if (lane >= off) incl += u; }
        if (lane == 63) sWave[wave] = incl;
        __syncthreads();
        int before = sCarry;
        for (int w = 0; w < wave; ++w) before += sWave[w];
        if (i < fw.n_blocks) { fw.start[i] = before + incl - v; fw.cursor[i] = before + incl - v; }
        __syncthreads();
        if (threadIdx.x == 1023) sCarry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) fw.start[fw.n_blocks] = sCarry;
}

// add one (sample, destination tile) contribution straight to the accumulator (fallback: no workspace / records overflow)
__device__ __forceinline__ void far_atomic_add(const NsffSplatArgs& a, const Landing& L, int s, const float (&src)[5], int only_tile,
                                               int tiles_x) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int cx = L.nwx + (k & 1), cy = L.nwy + (k >> 1);
        if (cx < 0 || cx >= a.W || cy < 0 || cy >= a.H) continue;
        if (only_tile >= 0 && (cy / TILE_Y) * tiles_x + cx / TILE_X != only_tile) continue;
        float* dst = a.accum + (((long long)cy * a.W + cx) * a.n_planes + s) * 8;
#pragma unroll
        for (int c = 0; c < 5; ++c) unsafeAtomicAdd(dst + c, src[c] * L.w[k]);
    }
}

// workgroups whose own samples include far ones: re-project them and append one record per destination tile
__global__ __launch_bounds__(256) void splat_bin_kernel(const NsffSplatArgs a) {
    __shared__ int sHist[MAX_TILES];      // records of this workgroup per destination tile: count, then slots handed out
    __shared__ int sBase[MAX_TILES];      // first record of this workgroup's range in that destination
    const FarWork fw = far_work(a);
    if (fw.count == nullptr) return;
    const SplatBlock blk_ = splat_block(fw.n_tiles, (a.n_planes + PL - 1) / PL);
    if (!blk_.live) return;
    const int blk = blk_.tile + fw.n_tiles * blk_.pg;
    if (fw.own[blk] == 0) return;                         // (the usual case for a trained flow field)
    const int S = a.n_planes;
    const int tx = blk_.tile % fw.tiles_x, ty = blk_.tile / fw.tiles_x;
    const int x0 = tx * TILE_X, y0 = ty * TILE_Y, s0 = blk_.pg * PL;
    for (int i = threadIdx.x; i < fw.n_tiles; i += 256) sHist[i] = 0;
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        for (int item = threadIdx.x; item < TILE_X * TILE_Y * PL; item += 256) {
            const int j = item % PL, rp = item / PL;
            const int px = x0 + rp % TILE_X, py = y0 + rp / TILE_X, sp = s0 + j;
            if (px >= a.W || py >= a.H || sp >= S) continue;
            const long long idx = ((long long)py * a.W + px) * S + sp;
            const Landing L = project_sample(a, idx, px, py);
            if (!L.finite || is_near(L, px, py)) continue;
            if (pass == 0) {
                for_each_dest_tile(a, L, fw.tiles_x, [&](int t) { atomicAdd(&sHist[t], 1); });
            } else {
                const float* cp = a.rgb + idx * 3;
                const float src[5] = {cp[0], cp[1], cp[2], a.alpha[idx], 1.0f};
                for_each_dest_tile(a, L, fw.tiles_x, [&](int t) {
                    const long long slot = (long long)sBase[t] + atomicAdd(&sHist[t], 1);
                    if (slot < fw.capacity) {
                        float4* r = reinterpret_cast<float4*>(fw.records + slot * 8);
                        r[0] = make_float4(L.ox, L.oy, __int_as_float(j), src[0]);
                        r[1] = make_float4(src[1], src[2], src[3], 0.f);
                    } else {                                 // workspace too small for this frame: the old route for the rest
                        far_atomic_add(a, L, sp, src, t, fw.tiles_x);
                    }
                });
            }
        }
        __syncthreads();
        if (pass == 0) {                                      // reserve this workgroup's range in every destination it feeds
            for (int t = threadIdx.x; t < fw.n_tiles; t += 256) {
                const int c = sHist[t];
                sBase[t] = c ? atomicAdd(fw.cursor + t + fw.n_tiles * blk_.pg, c) : 0;
                sHist[t] = 0;
            }
            __syncthreads();
        }
    }
}

// one workgroup per destination block: its records -> LDS accumulator -> added to the block splat_tiles_kernel wrote
__global__ __launch_bounds__(256) void splat_gather_kernel(const NsffSplatArgs a) {
    __shared__ double sAcc[TILE_X * TILE_Y * PL * 5];          // (double: see lds_add)
    const FarWork fw = far_work(a);
    if (fw.count == nullptr) return;
    const SplatBlock blk_ = splat_block(fw.n_tiles, (a.n_planes + PL - 1) / PL);
    if (!blk_.live) return;
    const int blk = blk_.tile + fw.n_tiles * blk_.pg;
    const long long first = fw.start[blk];
    long long n = fw.count[blk];
    if (first + n > fw.capacity) n = fw.capacity > first ? fw.capacity - first : 0;     // (the overflow went through atomics)
    if (n <= 0) return;
    const int S = a.n_planes;
    const int tx = blk_.tile % fw.tiles_x, ty = blk_.tile / fw.tiles_x;
    const int x0 = tx * TILE_X, y0 = ty * TILE_Y, s0 = blk_.pg * PL;
    for (int i = threadIdx.x; i < TILE_X * TILE_Y * PL * 5; i += 256) sAcc[i] = 0.0;
    __syncthreads();
    for (long long r = threadIdx.x; r < n; r += 256) {
        const float4* rec = reinterpret_cast<const float4*>(fw.records + (first + r) * 8);
        const float4 r0 = rec[0], r1 = rec[1];
        Landing L;
        L.ox = r0.x; L.oy = r0.y; L.finite = true;
        L.nwx = (int)floorf(r0.x); L.nwy = (int)floorf(r0.y);
        landing_weights(L);
        const int j = __float_as_int(r0.z);
        const float src[5] = {r0.w, r1.x, r1.y, r1.z, 1.0f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cx = L.nwx + (k & 1) - x0, cy = L.nwy + (k >> 1) - y0;
            if (cx < 0 || cx >= TILE_X || cy < 0 || cy >= TILE_Y) continue;
            if (L.nwx + (k & 1) >= a.W || L.nwy + (k >> 1) >= a.H) continue;
            double* dst = sAcc + ((cy * TILE_X + cx) * PL + j) * 5;
#pragma unroll
            for (int c = 0; c < 5; ++c) lds_add(dst + c, src[c] * L.w[k]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TILE_X * TILE_Y * PL; i += 256) {
        const int j = i % PL, pix = i / PL;
        const int px = x0 + pix % TILE_X, py = y0 + pix / TILE_X, sp = s0 + j;
        if (px >= a.W || py >= a.H || sp >= S) continue;
        const double* v = sAcc + i * 5;
        if (v[4] == 0.0) continue;                            // nothing landed in this cell
        float4* dst = reinterpret_cast<float4*>(a.accum + (((long long)py * a.W + px) * S + sp) * 8);
        float4 d0 = dst[0], d1 = dst[1];
        d0.x += (float)v[0]; d0.y += (float)v[1]; d0.z += (float)v[2]; d0.w += (float)v[3]; d1.x += (float)v[4];
        dst[0] = d0; dst[1] = d1;
    }
}

// Far samples without a workspace (or on frames of more than MAX_TILES blocks): device-scope atomics.
__global__ __launch_bounds__(256) void splat_far_kernel(const NsffSplatArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;       // (pixel, plane), plane fastest
    const long long total = (long long)a.H * a.W * a.n_planes;
    if (idx >= total) return;
    const int S = a.n_planes;
    const long long pix = idx / S;
    const int s = (int)(idx - pix * S);
    const int px = (int)(pix % a.W), py = (int)(pix / a.W);
    const Landing L = project_sample(a, idx, px, py);
    if (!L.finite || is_near(L, px, py)) return;
    const float* cp = a.rgb + idx * 3;
    const float src[5] = {cp[0], cp[1], cp[2], a.alpha[idx], 1.0f};
    far_atomic_add(a, L, s, src, -1, 0);
}

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void mpi_composite_kernel(const NsffMpiArgs a) {
    const int lane = threadIdx.x & 63;
    const long long pix = (long long)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (pix >= (long long)a.H * a.W) return;
    const int S = a.n_planes;
    const float dt = a.dt, omdt = 1.0f - a.dt;
    float carry = 1.0f;                       // transmittance 1 - A in front of this chunk
    float acc[4] = {0.f, 0.f, 0.f, 0.f};      // r, g, b, depth
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane;
        float c_rgb[3] = {0.f, 0.f, 0.f}, c_a = 0.f, z = 0.f;
        if (s < S) {
            const long long e = pix * S + s;
            const float4 f0 = *reinterpret_cast<const float4*>(a.accum_fw + e * 8);
            const float4 b0 = *reinterpret_cast<const float4*>(a.accum_bw + e * 8);
            float fn = a.accum_fw[e * 8 + 4], bn = a.accum_bw[e * 8 + 4];
            fn = fn == 0.0f ? 1.0f : fn;
            bn = bn == 0.0f ? 1.0f : bn;
            const float fa = f0.w / fn, ba = b0.w / bn;
            const float sa = a.static_alpha[e];
            const float* sr = a.static_rgb + e * 3;
            const float fr[3] = {f0.x / fn, f0.y / fn, f0.z / fn}, br[3] = {b0.x / bn, b0.y / bn, b0.z / bn};
#pragma unroll
            for (int c = 0; c < 3; ++c) c_rgb[c] = fr[c] * fa * omdt + br[c] * ba * dt + sr[c] * sa;
            c_a = 1.0f - (1.0f - (fa * omdt + ba * dt)) * (1.0f - sa);
            z = a.zs[e];
        }
        const float incl = wave_scan_mul(1.0f - c_a, lane);
        float excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 1.0f;
        const float T = carry * excl;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += T * c_rgb[c];
        acc[3] += T * c_a * z;
        carry *= __shfl(incl, 63);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = wave_sum(acc[c]);
    if (lane == 0) {
        a.rgb[pix * 3 + 0] = acc[0];
        a.rgb[pix * 3 + 1] = acc[1];
        a.rgb[pix * 3 + 2] = acc[2];
        a.depth[pix] = acc[3];
    }
}

}  // namespace

extern "C" {

int64_t nsff_splat_work_bytes(int32_t H, int32_t W, int32_t n_planes) {
    if (H < 1 || W < 1 || n_planes < 1) return -1;
    const long long tiles = (long long)((W + TILE_X - 1) / TILE_X) * ((H + TILE_Y - 1) / TILE_Y);
    const long long n_blocks = tiles * ((n_planes + PL - 1) / PL);
    const long long head = ((4 * n_blocks + 8) * 4 + 31) / 32 * 32;
    return head + 2LL * H * W * n_planes * 32;
}

int nsff_splat_planes(const NsffSplatArgs* args, void* stream) {
    if (!args) return NSFF_ERR_NULL;
    const NsffSplatArgs& a = *args;
    if (a.H < 1 || a.W < 1 || a.n_planes < 1) return NSFF_ERR_INVALID;
    if (!a.xyz || !a.flow || !a.rgb || !a.alpha || !a.accum) return NSFF_ERR_NULL;
    if (reinterpret_cast<uintptr_t>(a.accum) & 15) return NSFF_ERR_ALIGN;
    const long long total = (long long)a.H * a.W * a.n_planes;
    const unsigned tiles = (unsigned)(((a.W + TILE_X - 1) / TILE_X) * ((a.H + TILE_Y - 1) / TILE_Y));
    const long long n_blocks = (long long)tiles * ((a.n_planes + PL - 1) / PL);
    if (n_blocks > 0x7ffffff0LL) return NSFF_ERR_INVALID;
    const dim3 grid((unsigned)(SPLAT_XCD ? 8 * ((n_blocks + 7) / 8) : n_blocks));          // (see splat_block)
    hipStream_t st = (hipStream_t)stream;
    const long long head = ((4 * n_blocks + 8) * 4 + 31) / 32 * 32;
    const bool binned = a.work != nullptr && (int)tiles <= MAX_TILES && a.work_bytes >= head + 32 &&
                        !(reinterpret_cast<uintptr_t>(a.work) & 31);
    NsffSplatArgs k = a;
    if (!binned) { k.work = nullptr; k.work_bytes = 0; }
    else if (hipMemsetAsync(a.work, 0, (size_t)head, st) != hipSuccess) return nsff_launch_status();
    hipLaunchKernelGGL(splat_tiles_kernel, grid, dim3(256), 0, st, k);
    if (binned) {
        hipLaunchKernelGGL(splat_scan_kernel, dim3(1), dim3(1024), 0, st, k);
        hipLaunchKernelGGL(splat_bin_kernel, grid, dim3(256), 0, st, k);
        hipLaunchKernelGGL(splat_gather_kernel, grid, dim3(256), 0, st, k);
    } else {
        hipLaunchKernelGGL(splat_far_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, k);
    }
    return nsff_launch_status();
}

int nsff_mpi_composite(const NsffMpiArgs* args, void* stream) {
    if (!args) return NSFF_ERR_NULL;
    const NsffMpiArgs& a = *args;
    if (a.H < 1 || a.W < 1 || a.n_planes < 1) return NSFF_ERR_INVALID;
    if (!a.accum_fw || !a.accum_bw || !a.static_rgb || !a.static_alpha || !a.zs || !a.rgb || !a.depth) return NSFF_ERR_NULL;
    if ((reinterpret_cast<uintptr_t>(a.accum_fw) | reinterpret_cast<uintptr_t>(a.accum_bw)) & 15) return NSFF_ERR_ALIGN;
    const long long pixels = (long long)a.H * a.W;
    hipLaunchKernelGGL(mpi_composite_kernel, dim3((unsigned)((pixels + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK)),
                       dim3(64 * WAVES_PER_BLOCK), 0, (hipStream_t)stream, a);
    return nsff_launch_status();
}

}  // extern "C"
