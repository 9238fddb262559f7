// k_lk.h — pyramidal Lucas-Kanade tracker, one 64-lane wavefront per point.
// Replaces cv::calcOpticalFlowPyrLK (OpenCV lkpyramid.cpp LKTrackerInvoker) at
// reference src/frontend.cpp:353-357 (TrackLastFrame) and :105-109
// (FindFeaturesInRight).  Mirrors oracle/orc_image.c:lk_level operation for
// operation; integer patches and the exact-integer normal-equation sums make
// the result bit-exact against the oracle.
//
// Mapping: one wave per point, one wave per workgroup; window 11x11 = 121 pixels -> 2 pixels per lane.  Per level the wave
//   1. stages the 14 x 20 I neighbourhood (aligned dword loads from the
//      stored-border pyramid -> LDS, no index arithmetic),
//   2. computes the Scharr derivatives of the 12x12 inner positions on the fly,
//      3 horizontally adjacent positions per lane so the vertical [3 10 3] /
//      [-1 0 1] passes are shared (the reference materialises a full int16x2
//      derivative image per level — 5.3 bytes/pixel of HBM traffic never needed),
//   3. interpolates its two I/Ix/Iy samples (Q14 weights, v_dot2_i32_i16) into
//      registers and wave-reduces A11,A12,A22 (int32, exact),
//   4. stages a 32x32 J search region into LDS with aligned dword loads — stored EXPANDED, one dword per position
//      holding the tap pair (J[x] | J[x+1] << 16), so that a bilinear sample is one ds_read2_b32 (rows y, y+1) and two
//      v_dot2_i32_i16 — and iterates entirely out of LDS (re-staging only if the window leaves the region); b1, b2 are
//      one i16 dot product per lane each and are summed over the wave by v_permlane16_swap + ONE DPP row reduction
//      (int32; a 64-bit scalar tail for adversarial patches).  Round 5: 71 -> 57 VALU instructions per iteration.
// LDS discipline: every LDS access is a naturally aligned dword (pairs via
// ds_read2_b32); bytes at arbitrary offsets are extracted with v_perm_b32 /
// v_alignbyte_b32 using a per-lane selector.  A misaligned ds_read_u16/b32 is
// legal on gfx950 but is replayed lane by lane (measured ~40 LDS-pipe cycles
// per wave instruction, tools/ubench2.hip) and made an earlier version of this
// kernel LDS-pipe bound.  Beyond that the kernel is VALU-issue bound (the
// per-iteration arithmetic is mostly wave-uniform bookkeeping), so it uses 16-bit
// dot products for the bilinear taps, 24-bit multiplies and no 64-bit vector
// arithmetic.  All control flow is wave-uniform; waves never synchronise.
#pragma once
#include "dev_common.h"

struct LkJob { int prev_slot, next_slot, pt_ofs, npts; };
struct LkParams {
    int max_level;
    int max_count;
    double eps2;
    double min_eig_thr;
    int use_initial_flow;
    // smallest float x with (double)(x / (2 * 121)) >= min_eig_thr (the quotient is monotone in x), so the
    // min-eigenvalue test is one compare; eig_use_div = 1 keeps the division (threshold search did not settle)
    float eig_num_thr;
    int eig_use_div;
    // launch geometry (set by launch_lk): the grid is one-dimensional and deals the jobs over the 8 XCDs — the
    // hardware hands consecutive workgroup ids to consecutive XCDs, so job j runs on XCD j % 8 and all the points
    // of an image pair share one L2 (with a (points, jobs) grid every XCD fetched every pyramid)
    int njobs, blocks_per_job;
};

#define LK_WIN 11
#define LK_NPIX 121
#define LK_W_BITS 14
#define LK_REG 32
// Waves (= points) per workgroup.  The waves of this kernel never meet, but a workgroup keeps its wave slots, registers and LDS
// until its LAST wave is done, and the iteration counts have a heavy tail (a point that runs max_count iterations on every
// level takes five times the mean): with 4 waves per workgroup three finished waves' resources waited for the fourth.
// One wave per workgroup: 367 -> 325-335 us per 512 x 150 points temporal, 295 -> 275 stereo, the bench +2 % (round 5).
#ifndef LK_WAVES_PER_BLOCK
#define LK_WAVES_PER_BLOCK 1
#endif
#define LK_IROW 20            // bytes per staged I row (5 aligned dwords)

typedef uint32_t lk_u32_ua __attribute__((aligned(1)));
typedef uint16_t lk_u16_ua __attribute__((aligned(1)));
typedef short lk_s2 __attribute__((ext_vector_type(2)));

// a.lo*b.lo + a.hi*b.hi + c on signed 16-bit halves (v_dot2_i32_i16): one bilinear tap pair
__device__ __forceinline__ int lk_dot2(uint32_t a, uint32_t b, int c)
{
    // the three-source form: the compiler's choice (v_dot2c, accumulator == destination) costs a v_mov per tap
    // pair whenever the accumulator is a loop-invariant constant
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// Bytes (a, a+1) of an LDS byte array, any a, widened to two u16 halves: one aligned
// ds_read2_b32 + one v_perm_b32 with the per-lane selector lk_pair_sel(a).
#define LK_PAIR_SEL0 0x0c010c00u
__device__ __forceinline__ uint32_t lk_pair_sel(int a) { return LK_PAIR_SEL0 + (uint32_t)(a & 3) * 0x00010001u; }
__device__ __forceinline__ uint32_t lk_pair_at(const uint32_t *base32, int a, uint32_t sel)
{
    const uint32_t *q = base32 + (a >> 2);
    return __builtin_amdgcn_perm(q[1], q[0], sel);
}

// (int)floorf(x) in one VALU instruction (v_cvt_flr_i32_f32) instead of v_floor + v_cvt
__device__ __forceinline__ int lk_floor_i(float x)
{
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

struct LkW { uint32_t top, bot; };   // (w00 | w01 << 16), (w10 | w11 << 16); w11 may be -1 (kept signed)

__device__ __forceinline__ LkW lk_weights(float a, float b)
{
    // OpenCV: w = cvRound(x * (1 << W_BITS)) with x = (1-a)(1-b), a(1-b), (1-a)b and cvRound == round-half-even.
    // Scaling by 2^14 commutes with every f32 rounding here (no under- or overflow), so the scale goes onto a
    // first: A = 2^14 a (exact), 2^14 - A == fl(2^14 (1 - a)) == 2^14 fl(1 - a).  Round-half-even to an integer
    // in [0, 2^14] is one f32 add of 1.5 * 2^23: the integer sits in the low mantissa bits of the sum, and the
    // low 16 bits of the magic constant are zero, so the Q14 halves are packed straight from the float bits.
    // Packed f32 arithmetic (v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per instruction, each rounded exactly
    // like its scalar form).
    typedef float lk_f2 __attribute__((ext_vector_type(2)));
    const float S = (float)(1 << LK_W_BITS), M = 12582912.f;                  // M = 0x4B400000
    const float A = a * S, An = S - A, bn = 1.f - b;
    const lk_f2 p0 = (lk_f2){An, A} * (lk_f2){bn, bn} + (lk_f2){M, M};        // (An bn + M, A bn + M)
    const uint32_t b00 = __float_as_uint(p0.x), b01 = __float_as_uint(p0.y), b10 = __float_as_uint(An * b + M);
    // w11 = 2^14 - w00 - w01 - w10 (may be -1, kept signed in its 16-bit half)
    const uint32_t w11 = ((1u << LK_W_BITS) + 3u * 0x4B400000u) - (b00 + b01 + b10);
    LkW w;
    w.top = __builtin_amdgcn_perm(b01, b00, 0x05040100u);
    w.bot = __builtin_amdgcn_perm(w11, b10, 0x05040100u);
    return w;
}

// bilinear sample of a u8 patch held in LDS (dword view base32, byte offset a of the top-left
// tap, row stride in bytes a multiple of 4), Q5 output: CV_DESCALE(sum, W_BITS - 5)
__device__ __forceinline__ int lk_sample_u8(const uint32_t *base32, int a, int row_stride, LkW w)
{
    const uint32_t sel = lk_pair_sel(a);
    const uint32_t *q = base32 + (a >> 2);            // one address; the lower row is an immediate offset
    const int rs = row_stride >> 2;
    const uint32_t t = __builtin_amdgcn_perm(q[1], q[0], sel), b = __builtin_amdgcn_perm(q[rs + 1], q[rs], sel);
    return lk_dot2(t, w.top, lk_dot2(b, w.bot, 1 << (LK_W_BITS - 5 - 1))) >> (LK_W_BITS - 5);
}

// lk_sample_u8(...) - iv with the subtraction folded into the accumulator: the caller passes
// acc0 = (1 << 8) - (iv << 9); (x - 512 iv) >> 9 == (x >> 9) - iv for the arithmetic shift.
__device__ __forceinline__ int lk_sample_diff(const uint32_t *base32, int a, int row_stride, LkW w, int acc0)
{
    const uint32_t sel = lk_pair_sel(a);
    const uint32_t *q = base32 + (a >> 2);
    const int rs = row_stride >> 2;
    const uint32_t t = __builtin_amdgcn_perm(q[1], q[0], sel), b = __builtin_amdgcn_perm(q[rs + 1], q[rs], sel);
    return lk_dot2(t, w.top, lk_dot2(b, w.bot, acc0)) >> (LK_W_BITS - 5);
}

// J samples in the iteration loop come from the EXPANDED search region (round 5): one dword per pixel position holding the
// horizontal tap pair (J[x] | J[x + 1] << 16), so that a bilinear sample is ONE ds_read2_b32 (the pair of this row and of the
// row below, 32 dwords on) and two dot products — the byte region cost an address split, a selector and two v_perm per
// sample, 10 of the iteration's 78 VALU instructions, on every iteration; the expansion costs 20 per staging.
// addr: LDS byte address of the top-left tap's pair (per-lane constant + 4 x the window's offset in the region).
typedef const __attribute__((address_space(3))) uint32_t *lk_lds_u32;
__device__ __forceinline__ int lk_sample_diff_at(uint32_t addr, LkW w, int acc0)
{
    lk_lds_u32 q = (lk_lds_u32)(uintptr_t)addr;
    const uint32_t t = q[0], b = q[LK_REG];
    return lk_dot2(t, w.top, lk_dot2(b, w.bot, acc0)) >> (LK_W_BITS - 5);
}

// The 32x32 J search region around (cx, cy), one dword per lane and row octet: loads issued here ...
struct LkJRegs { uint32_t v[4]; };
// J0 = slot base, lofs = byte offset of the level's pixel (0, 0) inside the slot: all per-lane address
// arithmetic stays in 32 bits (uniform 64-bit base + unsigned 32-bit lane offset = the saddr form of global_load)
__device__ __forceinline__ LkJRegs lk_stage_J_issue(const uint8_t *J0, uint32_t lofs, int pitch, int w, int h,
                                                    int cx, int cy, int lane, int &rx0, int &ry0)
{
    rx0 = __builtin_amdgcn_readfirstlane((cx - 10) & ~3);
    ry0 = __builtin_amdgcn_readfirstlane(cy - 10);
    const int r = lane >> 3, c4 = lane & 7;
    LkJRegs o;
    if (rx0 >= -SVS_BORDER && rx0 + LK_REG <= w + SVS_BORDER && ry0 >= -SVS_BORDER && ry0 + LK_REG <= h + SVS_BORDER) {
        const uint32_t p = lofs + (uint32_t)((ry0 + r) * pitch + (rx0 + c4 * 4));
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o.v[k] = *reinterpret_cast<const uint32_t *>(J0 + (p + (uint32_t)(8 * k * pitch)));
    } else {
        const int gxmax = (w + SVS_BORDER - 4) & ~3;
        const int gx = max(-SVS_BORDER, min(rx0 + c4 * 4, gxmax));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int gy = max(-SVS_BORDER, min(ry0 + r + 8 * k, h + SVS_BORDER - 1));
            o.v[k] = *reinterpret_cast<const uint32_t *>(J0 + (lofs + (uint32_t)(gy * pitch + gx)));
        }
    }
    return o;
}
// ... and written to LDS here, so that the latency of the loads hides behind whatever sits in between.  The lane's dword
// (bytes x .. x + 3 of a row) becomes the four tap pairs (x, x+1) .. (x+3, x+4); the fifth byte is the first of the next lane's
// dword of the same row (row_shl:1 — the last lane of a row gets a foreign byte for the pair (31, 32), which no window reads:
// a window starts at offset <= 20 and is 11 wide, its right taps end at 31).
__device__ __forceinline__ void lk_stage_J_commit(uint32_t *sJ, const LkJRegs &o, int lane)
{
    typedef uint32_t lk_u4 __attribute__((ext_vector_type(4)));
    const int r = lane >> 3, c4 = lane & 7;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t cur = o.v[k];
        const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cur, 0x101 /* row_shl:1 */, 0xf, 0xf, true);
        lk_u4 pr;
        pr.x = __builtin_amdgcn_perm(nxt, cur, 0x0c010c00u);
        pr.y = __builtin_amdgcn_perm(nxt, cur, 0x0c020c01u);
        pr.z = __builtin_amdgcn_perm(nxt, cur, 0x0c030c02u);
        pr.w = __builtin_amdgcn_perm(nxt, cur, 0x0c040c03u);
        *reinterpret_cast<lk_u4 *>(sJ + (r + 8 * k) * LK_REG + 4 * c4) = pr;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void lk_stage_J(uint32_t *sJ, const uint8_t *J0, uint32_t lofs, int pitch, int w, int h,
                                           int cx, int cy, int lane, int &rx0, int &ry0)
{
    const LkJRegs o = lk_stage_J_issue(J0, lofs, pitch, w, h, cx, cy, lane, rx0, ry0);
    lk_stage_J_commit(sJ, o, lane);
}

// (float)s * scale for a wave-uniform 64-bit sum, |s| < 2^32 (121 x 8160 x 4080), scale a power of
// two.  The sums fit int32 except for adversarial patches; beyond that the value is halved with a
// sticky low bit, which rounds to the same float (the rounding position is far above bit 1), and
// the factor 2 moves into the scale.  Both cases are integer selects on the scalar unit plus one
// v_cvt_f32_i32 and one multiply — no 64-bit or f64 vector arithmetic in the iteration loop.
__device__ __forceinline__ float lk_sum_to_f32(long long s, float scale)
{
    const int lo = (int)s;
    const bool fits = (long long)lo == s;
    const int t = fits ? lo : (int)((s >> 1) | (s & 1));
    const uint32_t kb = fits ? __float_as_uint(scale) : __float_as_uint(2.f * scale);   // integer select: stays scalar
    return (float)t * __uint_as_float(kb);
}

// Wave totals of two (four) int32 lane values whose sums of absolute values fit int32 (round 5: gfx950's row swaps).
// v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of its second, so
// (a, b) -> ([a0 b0 a2 b2], [a1 b1 a3 b3]) and one add leaves the row pairs' sums of a in rows 0 / 2 and of b in rows 1 / 3;
// v_permlane32_swap does the same with the 32-lane halves: a second add leaves, lane by lane, the column sums of a in row 0
// and of b in row 1 (of c, d in rows 2, 3).  ONE 4-step DPP row reduction then finishes all of them at once and one
// v_readlane per sum fetches it — 14 VALU instructions instead of 28 for four sums; for two sums the second swap is
// replaced by two more v_readlane and two scalar additions (10 instead of 14).  Integer sums, exact in any order.
typedef unsigned lk_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int lk_row_reduce(int w)
{
    w += dpp_i32<SVS_DPP_XOR1>(w);
    w += dpp_i32<SVS_DPP_XOR2>(w);
    w += dpp_i32<SVS_DPP_HALF_MIRROR>(w);
    w += dpp_i32<SVS_DPP_MIRROR>(w);
    return w;
}
__device__ __forceinline__ void lk_wave_sum2_i32(int a, int b, int &sa, int &sb)
{
    const lk_u2 r = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false);
    const int v = (int)(r.x + r.y);
    // rows [a01 b01 a23 b23]: the row reduction, then the two halves meet on the scalar unit (wrap-around is harmless: the
    // totals fit) — 10 VALU instructions
    const int w = lk_row_reduce(v);
    sa = __builtin_amdgcn_readlane(w, 0) + __builtin_amdgcn_readlane(w, 32);
    sb = __builtin_amdgcn_readlane(w, 16) + __builtin_amdgcn_readlane(w, 48);
}
__device__ __forceinline__ void lk_wave_sum4_i32(int a, int b, int c, int d, int &sa, int &sb, int &sc, int &sd)
{
    const lk_u2 r1 = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false);
    const lk_u2 r2 = __builtin_amdgcn_permlane16_swap((unsigned)c, (unsigned)d, false, false);
    const int v1 = (int)(r1.x + r1.y), v2 = (int)(r2.x + r2.y);           // rows [a01 b01 a23 b23], [c01 d01 c23 d23]
    const lk_u2 q = __builtin_amdgcn_permlane32_swap((unsigned)v1, (unsigned)v2, false, false);
    const int w = lk_row_reduce((int)(q.x + q.y));                          // rows: a, b, c, d
    sa = __builtin_amdgcn_readlane(w, 0);
    sb = __builtin_amdgcn_readlane(w, 16);
    sc = __builtin_amdgcn_readlane(w, 32);
    sd = __builtin_amdgcn_readlane(w, 48);
}

#ifndef LK_OCC_TEST
#define LK_OCC_TEST 0       // development: 1 = force 8 waves per SIMD, 2 = pad LDS down to 4 waves per SIMD
#endif
#if LK_OCC_TEST == 1
#define LK_OCC_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))
#else
#define LK_OCC_ATTR
#endif
__global__ void __launch_bounds__(64 * LK_WAVES_PER_BLOCK) LK_OCC_ATTR
k_lk(const LkJob *jobs, const uint8_t *pyr, PyrGeom g, const float2 *prev_xy, float2 *next_xy,
     uint8_t *status, float *err, LkParams prm)
{
    __shared__ uint32_t sI_all[LK_WAVES_PER_BLOCK][16 * LK_IROW / 4 + 2];     // 14 rows used, 16 staged
    __shared__ uint32_t sD_all[LK_WAVES_PER_BLOCK][144];
    __shared__ __attribute__((aligned(16))) uint32_t sJ_all[LK_WAVES_PER_BLOCK][LK_REG * LK_REG];     // expanded: one tap pair per position

#if LK_OCC_TEST == 2
    __shared__ uint32_t sPad[8192];
    sPad[threadIdx.x] = threadIdx.x;
    if (prm.max_count < 0) status[0] = (uint8_t)sPad[(threadIdx.x * 37) & 8191];
#endif
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = blockIdx.x >> 3, round = slot / prm.blocks_per_job;
    const int job = round * 8 + (blockIdx.x & 7);
    if (job >= prm.njobs) return;
    const LkJob jb = jobs[job];
    const int pi = (slot - round * prm.blocks_per_job) * LK_WAVES_PER_BLOCK + wave;
    if (pi >= jb.npts) return;
    const int pt = jb.pt_ofs + pi;
    uint32_t *sI32 = sI_all[wave];
    uint32_t *sD = sD_all[wave];
    uint32_t *sJ = sJ_all[wave];

    const uint8_t *slotI = pyr + (size_t)jb.prev_slot * g.slot_bytes;
    const uint8_t *slotJ = pyr + (size_t)jb.next_slot * g.slot_bytes;

    const float2 prevp = prev_xy[pt];
    float2 nextp = next_xy[pt];
    bool st = true;
    float errv = 0.f;

    // this lane's two window pixels (lanes >= 57 have only one; their second pixel is
    // aliased to pixel 0 and its I/Ix/Iy forced to zero so it drops out of every sum)
    const int p0 = lane, p1 = lane + 64;
    const bool has1 = p1 < LK_NPIX;
    const int wy0 = p0 / LK_WIN, wx0 = p0 - wy0 * LK_WIN;
    const int wy1 = has1 ? p1 / LK_WIN : 0, wx1 = has1 ? p1 - wy1 * LK_WIN : 0;
    const int oI0 = (wy0 + 1) * LK_IROW + wx0 + 1, oI1 = (wy1 + 1) * LK_IROW + wx1 + 1;
    const int oD0 = wy0 * 12 + wx0, oD1 = wy1 * 12 + wx1;
    const int oJ0 = wy0 * LK_REG + wx0, oJ1 = wy1 * LK_REG + wx1;
    // LDS byte addresses of the tap pairs of this lane's two window pixels at search-region offset 0
    const uint32_t aJ0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)sJ + 4u * (uint32_t)oJ0,
                   aJ1 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)sJ + 4u * (uint32_t)oJ1;
    // Scharr work split: lanes 0..47 -> row lane/4, columns 3*(lane%4) .. +2
    const int sr = lane >> 2, sc = (lane & 3) * 3;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    // f32 brackets of the f64 termination test dx^2 + dy^2 <= eps2: the f32 evaluation is within 2^-22
    // relative of the f64 one, so outside (e_lo, e_hi) it decides; inside, the f64 test runs
    const float e_lo = (float)(prm.eps2 * (1.0 - 2e-6)), e_hi = (float)(prm.eps2 * (1.0 + 2e-6));

    int max_level = prm.max_level;
    if (max_level > g.nlevels - 1) max_level = g.nlevels - 1;

    for (int level = max_level; level >= 0; --level) {
        const int w = g.w[level], h = g.h[level], pitch = g.pitch[level];
        // pixel (0, 0) of the level as a 32-bit offset into the slot (both slots share the geometry)
        const uint32_t lofs = (uint32_t)g.ofs[level] + (uint32_t)(SVS_BORDER * pitch + SVS_BORDER);
        const float lscale = __uint_as_float((uint32_t)(127 - level) << 23);     // 2^-level
        float px = prevp.x * lscale, py = prevp.y * lscale;
        float nx, ny;
        if (level == max_level) {
            if (prm.use_initial_flow) { nx = nextp.x * lscale; ny = nextp.y * lscale; }
            else { nx = px; ny = py; }
        } else { nx = nextp.x * 2.f; ny = nextp.y * 2.f; }
        nextp.x = nx; nextp.y = ny;

        px -= 5.f; py -= 5.f;
        const int ipx = __builtin_amdgcn_readfirstlane(lk_floor_i(px)), ipy = __builtin_amdgcn_readfirstlane(lk_floor_i(py));
        const uint32_t wext = (uint32_t)(w + LK_WIN), hext = (uint32_t)(h + LK_WIN);     // -11 <= v < len as one unsigned compare
        if ((uint32_t)(ipx + LK_WIN) >= wext || (uint32_t)(ipy + LK_WIN) >= hext) {
            if (level == 0) { st = false; errv = 0.f; }
            continue;
        }
        const LkW iw = lk_weights(px - (float)ipx, py - (float)ipy);

        // 1. stage I neighbourhood: rows ipy-1..ipy+12, 20 bytes from the aligned column xs <= ipx-1
        const int xs = (ipx - 1) & ~3, a0 = (ipx - 1) & 3;   // patch byte (r, c) lives at r*LK_IROW + a0 + c
        __builtin_amdgcn_wave_barrier();
        // every lane loads (16 rows x 4 dwords, then the fifth dword of row lane % 16: rows 14, 15 and the
        // duplicates are never read) — no exec masking; the J region of this level's start position follows
        // at once so that both load latencies overlap
        nx -= 5.f; ny -= 5.f;
        int rx0, ry0;
        LkJRegs jr;
        {
            const uint32_t ib = lofs + (uint32_t)((ipy - 1) * pitch + xs);     // >= 0: the stored border covers rows >= -16
            const uint32_t v0 = *reinterpret_cast<const uint32_t *>(slotI + (ib + (uint32_t)((lane >> 2) * pitch + (lane & 3) * 4)));
            const uint32_t v1 = *reinterpret_cast<const uint32_t *>(slotI + (ib + (uint32_t)((lane & 15) * pitch + 16)));
            jr = lk_stage_J_issue(slotJ, lofs, pitch, w, h, lk_floor_i(nx), lk_floor_i(ny), lane, rx0, ry0);
            sI32[(lane >> 2) * (LK_IROW / 4) + (lane & 3)] = v0;
            sI32[(lane & 15) * (LK_IROW / 4) + 4] = v1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 2. Scharr at the 12x12 positions (ipx+c, ipy+r); zero outside the image.  Packed 16-bit arithmetic: every
        // intermediate fits 16 bits (|t0| <= 4080, |t1| <= 255, |dx|, |dy| <= 8160 < 2^15) and v_pk_add / v_pk_mad /
        // v_pk_mul_lo / v_pk_sub issue in the same 4-cycle class as the 32-bit v_mad_i32_i24 they replace, two results
        // each (tools/ubench_valu.hip, profiles/r3_ubench_valu_issue_rates.txt): ~35 instead of ~65 instructions.
        if (lane < 48) {
            const int ab = sr * LK_IROW + a0 + sc;      // top-left byte of the 3x5 block
            const int k8 = (ab & 3) * 8;
            const uint32_t *q = sI32 + (ab >> 2);
            const uint32_t al = q[0], ah = q[1], bl = q[LK_IROW / 4], bh = q[LK_IROW / 4 + 1],
                           cl = q[2 * (LK_IROW / 4)], ch = q[2 * (LK_IROW / 4) + 1];
            const uint32_t a4 = __builtin_amdgcn_alignbyte(ah, al, ab & 3), b4 = __builtin_amdgcn_alignbyte(bh, bl, ab & 3),
                           c4 = __builtin_amdgcn_alignbyte(ch, cl, ab & 3);
            // bytes (0, 1), (2, 3), (4, -) of each row as u16 pairs
            auto pr = [](uint32_t v) -> lk_s2 { return __builtin_bit_cast(lk_s2, v); };
            const lk_s2 A01 = pr(__builtin_amdgcn_perm(0u, a4, 0x0c010c00u)), A23 = pr(__builtin_amdgcn_perm(0u, a4, 0x0c030c02u)), A4 = pr((ah >> k8) & 0xffu);
            const lk_s2 B01 = pr(__builtin_amdgcn_perm(0u, b4, 0x0c010c00u)), B23 = pr(__builtin_amdgcn_perm(0u, b4, 0x0c030c02u)), B4 = pr((bh >> k8) & 0xffu);
            const lk_s2 C01 = pr(__builtin_amdgcn_perm(0u, c4, 0x0c010c00u)), C23 = pr(__builtin_amdgcn_perm(0u, c4, 0x0c030c02u)), C4 = pr((ch >> k8) & 0xffu);
            const lk_s2 k3 = { 3, 3 }, k10 = { 10, 10 };
            // t0 = (a + c) 3 + 10 b (vertical [3 10 3]), t1 = c - a (vertical [-1 0 1]) for the five columns
            const lk_s2 T001 = (A01 + C01) * k3 + B01 * k10, T023 = (A23 + C23) * k3 + B23 * k10, T04 = (A4 + C4) * k3 + B4 * k10;
            const lk_s2 T101 = C01 - A01, T123 = C23 - A23, T14 = C4 - A4;
            // dx_k = t0_{k+2} - t0_k; dy_k = (t1_{k+2} + t1_k) 3 + 10 t1_{k+1}
            const lk_s2 DX01 = T023 - T001, DX2 = T04 - T023;
            const lk_s2 M01 = pr(__builtin_amdgcn_alignbyte(__builtin_bit_cast(uint32_t, T123), __builtin_bit_cast(uint32_t, T101), 2));   // (t1_1, t1_2)
            const lk_s2 M2 = pr(__builtin_amdgcn_alignbyte(__builtin_bit_cast(uint32_t, T14), __builtin_bit_cast(uint32_t, T123), 2));     // (t1_3, t1_4)
            const lk_s2 DY01 = (T123 + T101) * k3 + M01 * k10, DY2 = (T14 + T123) * k3 + M2 * k10;
            const uint32_t dx01 = __builtin_bit_cast(uint32_t, DX01), dy01 = __builtin_bit_cast(uint32_t, DY01);
            const uint32_t o0 = __builtin_amdgcn_perm(dy01, dx01, 0x05040100u), o1 = __builtin_amdgcn_perm(dy01, dx01, 0x07060302u),
                           o2 = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, DY2), __builtin_bit_cast(uint32_t, DX2), 0x05040100u);
            const bool rowin = (uint32_t)(ipy + sr) < (uint32_t)h;
            // zero outside the image: a select on the packed pair, no divergent branch
            sD[sr * 12 + sc + 0] = o0 & (0u - (uint32_t)(rowin & ((uint32_t)(ipx + sc + 0) < (uint32_t)w)));
            sD[sr * 12 + sc + 1] = o1 & (0u - (uint32_t)(rowin & ((uint32_t)(ipx + sc + 1) < (uint32_t)w)));
            sD[sr * 12 + sc + 2] = o2 & (0u - (uint32_t)(rowin & ((uint32_t)(ipx + sc + 2) < (uint32_t)w)));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // 3. interpolate the two samples of this lane
        int iv0, ix0, iy0, iv1, ix1, iy1;
        {
            iv0 = lk_sample_u8(sI32, oI0 + a0, LK_IROW, iw);
            const uint32_t d00 = sD[oD0], d01 = sD[oD0 + 1], d10 = sD[oD0 + 12], d11 = sD[oD0 + 13];
            ix0 = lk_dot2(__builtin_amdgcn_perm(d01, d00, 0x05040100u), iw.top,
                          lk_dot2(__builtin_amdgcn_perm(d11, d10, 0x05040100u), iw.bot, 1 << (LK_W_BITS - 1))) >> LK_W_BITS;
            iy0 = lk_dot2(__builtin_amdgcn_perm(d01, d00, 0x07060302u), iw.top,
                          lk_dot2(__builtin_amdgcn_perm(d11, d10, 0x07060302u), iw.bot, 1 << (LK_W_BITS - 1))) >> LK_W_BITS;
        }
        {
            iv1 = lk_sample_u8(sI32, oI1 + a0, LK_IROW, iw);
            const uint32_t d00 = sD[oD1], d01 = sD[oD1 + 1], d10 = sD[oD1 + 12], d11 = sD[oD1 + 13];
            ix1 = lk_dot2(__builtin_amdgcn_perm(d01, d00, 0x05040100u), iw.top,
                          lk_dot2(__builtin_amdgcn_perm(d11, d10, 0x05040100u), iw.bot, 1 << (LK_W_BITS - 1))) >> LK_W_BITS;
            iy1 = lk_dot2(__builtin_amdgcn_perm(d01, d00, 0x07060302u), iw.top,
                          lk_dot2(__builtin_amdgcn_perm(d11, d10, 0x07060302u), iw.bot, 1 << (LK_W_BITS - 1))) >> LK_W_BITS;
            if (!has1) { ix1 = 0; iy1 = 0; }
        }
        // A sums (exact: 121 x 4080^2 < 2^31); the fourth lane of the butterfly is idle
        int sA11, sA12, sA22, sIdle;
        lk_wave_sum4_i32(__mul24(ix0, ix0) + __mul24(ix1, ix1), __mul24(ix0, iy0) + __mul24(ix1, iy1), __mul24(iy0, iy0) + __mul24(iy1, iy1), 0,
                         sA11, sA12, sA22, sIdle);
        const float A11 = (float)sA11 * FLT_SCALE, A12 = (float)sA12 * FLT_SCALE, A22 = (float)sA22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float dd = A11 - A22;
        // The min-eigenvalue test is a comparison: the hardware square root (v_sqrt_f32, <= 1 ulp) decides unless the
        // numerator lands within 4 ulp(sqrt) of the threshold; only then the correctly rounded sqrtf (16 instructions
        // of fix-up) is evaluated — the decision is the oracle's bit for bit either way.
        const float q2 = dd * dd + 4.f * A12 * A12;
        float eigNum;
        if (prm.eig_use_div) eigNum = A22 + A11 - sqrtf(q2);
        else {
            const float sq = __builtin_amdgcn_sqrtf(q2);
            eigNum = A22 + A11 - sq;
            if (fabsf(eigNum - prm.eig_num_thr) <= sq * 4.8e-7f + 1e-30f) eigNum = A22 + A11 - sqrtf(q2);
        }
        const bool eigSmall = prm.eig_use_div ? (double)(eigNum / (float)(2 * LK_WIN * LK_WIN)) < prm.min_eig_thr : eigNum < prm.eig_num_thr;
        if (eigSmall || D < 1.1920928955078125e-07f) {
            if (level == 0) st = false;
            continue;
        }
        D = 1.f / D;
        // |d| <= 8160 (Q5 difference of two u8 interpolations), so |b| and all its partial sums stay below
        // 8160 * sum|Ix| <= 8160 * 11 * sqrt(sum Ix^2) (Cauchy-Schwarz over the 121 pixels): with sum Ix^2 and
        // sum Iy^2 <= 5.7e8 that is < 2^31 and the b sums fit int32 (always, short of adversarial patches)
        const bool narrow = sA11 <= 570000000 && sA22 <= 570000000;
        const int acc00 = (1 << (LK_W_BITS - 5 - 1)) - (iv0 << (LK_W_BITS - 5)), acc01 = (1 << (LK_W_BITS - 5 - 1)) - (iv1 << (LK_W_BITS - 5));
        // (Ix0 | Ix1 << 16), (Iy0 | Iy1 << 16): b's per-lane terms are one i16 dot product each
        const uint32_t ixx = __builtin_amdgcn_perm((uint32_t)ix1, (uint32_t)ix0, 0x05040100u), iyy = __builtin_amdgcn_perm((uint32_t)iy1, (uint32_t)iy0, 0x05040100u);
        float pdx = 0.f, pdy = 0.f, dx = 0.f, dy = 0.f;
        const float Ds = D * FLT_SCALE;
        bool moved = false, osc = false;
        lk_stage_J_commit(sJ, jr, lane);

        for (int j = 0; j < prm.max_count; ++j) {
            const int inx = __builtin_amdgcn_readfirstlane(lk_floor_i(nx)), iny = __builtin_amdgcn_readfirstlane(lk_floor_i(ny));
            if ((uint32_t)(inx + LK_WIN) >= wext || (uint32_t)(iny + LK_WIN) >= hext) {
                if (level == 0) st = false;
                break;
            }
            int ox = inx - rx0, oy = iny - ry0;
            if ((uint32_t)ox > LK_REG - 12 || (uint32_t)oy > LK_REG - 12) {
                lk_stage_J(sJ, slotJ, lofs, pitch, w, h, inx, iny, lane, rx0, ry0);
                ox = inx - rx0; oy = iny - ry0;
            }
            const LkW jw = lk_weights(nx - (float)inx, ny - (float)iny);
            const uint32_t jo = 4u * (uint32_t)(oy * LK_REG + ox);
            const int d0 = lk_sample_diff_at(aJ0 + jo, jw, acc00);
            const int d1 = lk_sample_diff_at(aJ1 + jo, jw, acc01);
            // |d| < 2^14, |Ix|,|Iy| < 2^13: 24-bit multiplies, 16-lane row sums fit int32
            const uint32_t dd01 = __builtin_amdgcn_perm((uint32_t)d1, (uint32_t)d0, 0x05040100u);
            const int pb1 = lk_dot2(dd01, ixx, 0), pb2 = lk_dot2(dd01, iyy, 0);
            float b1, b2;
            if (narrow) {
                int s1, s2;
                lk_wave_sum2_i32(pb1, pb2, s1, s2);
                b1 = (float)s1; b2 = (float)s2;
            } else {
                b1 = lk_sum_to_f32(wave_sum_i32_wide(pb1), 1.f); b2 = lk_sum_to_f32(wave_sum_i32_wide(pb2), 1.f);
            }
            // b's 2^-20 rides on 1 / det (Ds): a power of two commutes with every rounding of this expression (no
            // operand comes near the subnormals: |b| >= 1 or 0, 2^-20 <= |A| < 2^10, 3e-6 < 1 / det < 8.4e6)
            dx = (A12 * b2 - A22 * b1) * Ds;
            dy = (A12 * b1 - A11 * b2) * Ds;
            nx += dx; ny += dy;
            moved = true;
            const float dd2 = dx * dx + dy * dy;
            if (dd2 < e_lo) break;
            if (dd2 <= e_hi && (double)dx * (double)dx + (double)dy * (double)dy <= prm.eps2) break;
            // (double)|v| < 0.01  <=>  |v| <= 0.01f  (0.01f is the largest float below 0.01)
            if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {
                osc = true;
                break;
            }
            pdx = dx; pdy = dy;
        }
        // the estimate leaves the loop in window-corner coordinates (the +5 is taken once, not per iteration)
        if (moved) {
            nextp.x = nx + 5.f; nextp.y = ny + 5.f;
            if (osc) { nextp.x -= dx * 0.5f; nextp.y -= dy * 0.5f; }
        }

        if (st && level == 0) {
            // level-0 residual ("err" output); can still clear status
            const float fx = nextp.x - 5.f, fy = nextp.y - 5.f;
            const int inx = __builtin_amdgcn_readfirstlane(lk_floor_i(fx)), iny = __builtin_amdgcn_readfirstlane(lk_floor_i(fy));
            if (inx < -LK_WIN || inx >= w || iny < -LK_WIN || iny >= h) {
                st = false;
                continue;
            }
            int ox = inx - rx0, oy = iny - ry0;
            if (ox < 0 || ox > LK_REG - 12 || oy < 0 || oy > LK_REG - 12) {
                lk_stage_J(sJ, slotJ, lofs, pitch, w, h, inx, iny, lane, rx0, ry0);
                ox = inx - rx0; oy = iny - ry0;
            }
            const LkW jw = lk_weights(fx - (float)inx, fy - (float)iny);
            const uint32_t jo = 4u * (uint32_t)(oy * LK_REG + ox);
            // (lk_sample_u8's rounding constant in the accumulator; the same taps from the expanded region)
            const int d0 = lk_sample_diff_at(aJ0 + jo, jw, 1 << (LK_W_BITS - 5 - 1)) - iv0;
            const int d1 = lk_sample_diff_at(aJ1 + jo, jw, 1 << (LK_W_BITS - 5 - 1)) - iv1;
            const int e = (d0 < 0 ? -d0 : d0) + (has1 ? (d1 < 0 ? -d1 : d1) : 0);
            const int serr = wave_sum_i32(e);
            errv = (float)serr * 1.f / (float)(32 * LK_WIN * LK_WIN);
        }
    }
    if (lane == 0) {
        next_xy[pt] = nextp;
        status[pt] = st ? 1 : 0;
        if (err) err[pt] = errv;
    }
}
