// slide_common.hpp — work decomposition shared by the register-resident sliding-window kernels
// (gauss_slide.hip, sobel_slide.hip, pipe_slide.hip).
//
// A work item is one wave walking one (frame, band of rows, strip of <= 62 lanes + 2 halo lanes).
// Band height trades two costs measured on MI355X:
//   * every band re-reads (and, for the Gaussian stages, re-computes) its warm-up rows, so tall bands are
//     cheaper per pixel;
//   * a launch should hold roughly ten "rounds" of resident waves, otherwise ramp-up and the last, partly
//     filled round show (64 x 4K frames in 128-row bands are only 4.25 rounds).
// So the height adapts to the batch: rows = frame rows x strips x frames / (10 x resident waves), clamped
// to [rows_min, rows_max].  In addition the last ~10 % of every frame is cut into short bands that are
// dispatched after all tall ones (blocks start in blockIdx order), so the tail of the launch is made of
// short items (+6 % on 64-frame batches at steady clocks).
#pragma once
#include <cstdint>
#include <cstdlib>

#include "common.hpp"

namespace mi355 {

constexpr int kSlideLanesOutMax = 62;  // 64 lanes minus one halo lane per side
constexpr int kSlideWavesPerBlock = 4;

// Rows [0, y_split) of every frame: nbands_a bands of rows_a rows (phase A, dispatched first);
// rows [y_split, h): nbands_b bands of rows_b rows (phase B, dispatched last).
struct BandPlan {
    int rows_a, nbands_a, rows_b, nbands_b, y_split;
    uint32_t nwork_a, nwork_b, nblocks_a, nblocks_b;
};

struct StripPlan {
    int quads, nstrips, lanes_out;
};

// lanes_pref > 0: strips of exactly that many output lanes whenever the row needs more than one strip (the
// kernels with one output BYTE per pixel want the 4-byte-per-lane stores of a strip to start on 64-byte
// boundaries: Sobel on 4K frames, same box: 60 lanes 5.17 TB/s, 56: 5.18, 52: 4.92, 48: 5.40, 44: 4.79, 40: 4.91)
inline StripPlan make_strip_plan(int w, int lanes_pref = 0)
{
    StripPlan s;
    s.quads = (w + 3) / 4;  // a ragged last quad counts
    s.nstrips = (s.quads + kSlideLanesOutMax - 1) / kSlideLanesOutMax;
    s.lanes_out = (s.quads + s.nstrips - 1) / s.nstrips;  // e.g. 4K: 960 quads = 16 strips x 60 lanes
    if (lanes_pref > 0 && lanes_pref <= kSlideLanesOutMax && s.nstrips > 1) {
        s.lanes_out = lanes_pref;
        s.nstrips = (s.quads + lanes_pref - 1) / lanes_pref;
    }
    static const int tune_lanes = [] {  // MI355_TUNE_LANES_OUT: tuning sweeps only
        const char* e = tune_env("MI355_TUNE_LANES_OUT");
        return e ? atoi(e) : 0;
    }();
    if (tune_lanes > 0 && tune_lanes <= kSlideLanesOutMax) {
        s.lanes_out = tune_lanes;
        s.nstrips = (s.quads + tune_lanes - 1) / tune_lanes;
    }
    return s;
}

// waves_per_simd: occupancy of the kernel (from its VGPR count); rows_min/rows_max: clamp of the tall bands
// rows_small: the band height is halved, down to this floor, while the launch has fewer than ~2800 work items —
// a launch too small to occupy the chip (one or a few frames) is cut finer.  Measured on ONE 4K frame: Gaussian
// k = 5 24-row bands 19 us, 12-row bands 16 us; k = 7 96 rows 54 us, 24 rows 26 us; k = 17 128 rows 95 us, 32 rows
// 55 us; Sobel 16 -> 8 rows +6 %; pipeline 24 -> 12 rows +17 %.  The floor is about twice the warm-up rows.
inline bool make_band_plan(int h, int nstrips, int nframes, int waves_per_simd, int rows_min, int rows_max,
                           int rows_tail, double tail_frac, int rows_small, BandPlan* out)
{
    const double resident = 256.0 * 4.0 * waves_per_simd;
    double rows = (double)h * nstrips * nframes / (10.0 * resident);
    int rows_big = (int)(rows < rows_min ? rows_min : (rows > rows_max ? rows_max : rows));
    // MI355_TUNE_* environment overrides exist for tuning sweeps only (tools/sweep*.sh); read once
    static const struct Tune {
        int band_rows = 0, tail_rows = 0;
        double tail_frac = -1.0;
        Tune()
        {
            if (const char* e = tune_env("MI355_TUNE_BAND_ROWS"))
                band_rows = atoi(e);
            if (const char* e = tune_env("MI355_TUNE_TAIL_ROWS"))
                tail_rows = atoi(e);
            if (const char* e = tune_env("MI355_TUNE_TAIL_FRAC"))
                tail_frac = atof(e);
        }
    } tune;
    while (rows_small > 0 && rows_big / 2 >= rows_small &&
           (size_t)nstrips * ((h + rows_big - 1) / rows_big) * nframes < 2800)
        rows_big /= 2;
    if (tune.band_rows > 0)
        rows_big = tune.band_rows;
    if (tune.tail_rows > 0)
        rows_tail = tune.tail_rows;
    if (tune.tail_frac >= 0.0)
        tail_frac = tune.tail_frac;
    BandPlan p{};
    int h_b = (int)(h * tail_frac);
    if (h_b < rows_tail || h - h_b < rows_big || tail_frac <= 0.0 || rows_tail >= rows_big)
        h_b = 0;  // small images: one phase
    const int h_a = h - h_b;
    p.nbands_a = (h_a + rows_big - 1) / rows_big;
    p.rows_a = (h_a + p.nbands_a - 1) / p.nbands_a;  // balance the bands
    p.y_split = h_a;
    if (h_b > 0) {
        p.nbands_b = (h_b + rows_tail - 1) / rows_tail;
        p.rows_b = (h_b + p.nbands_b - 1) / p.nbands_b;
    } else {
        p.nbands_b = 0;
        p.rows_b = 1;
    }
    const size_t na = (size_t)nstrips * p.nbands_a * nframes, nb = (size_t)nstrips * p.nbands_b * nframes;
    if (na + nb > 0x3FFFFFFFull)
        return false;
    p.nwork_a = (uint32_t)na;
    p.nwork_b = (uint32_t)nb;
    p.nblocks_a = (p.nwork_a + kSlideWavesPerBlock - 1) / kSlideWavesPerBlock;
    p.nblocks_b = (p.nwork_b + kSlideWavesPerBlock - 1) / kSlideWavesPerBlock;
    *out = p;
    return true;
}

#ifdef __HIPCC__
// Pins a wave-uniform row pointer in an SGPR pair.  Without it hipcc reassociates `frame + y * row_bytes +
// lane_offset` into (frame + lane_offset) + y * row_bytes and spends a v_mov + v_mad_u64_u32 + v_add per row
// access; with it the access uses the `saddr + 32-bit voffset` form and the row address costs no VALU at all.
template <typename T>
using global_ptr = __attribute__((address_space(1))) T*;  // keeps global_* (not flat_*) instructions

template <typename T>
__device__ __forceinline__ global_ptr<T> uniform_ptr(T* p)
{
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (global_ptr<T>)(((uint64_t)hi << 32) | lo);
}

// The saddr form is only selected when the 32-bit -> 64-bit extension of the lane offset happens in the basic
// block of the access (instruction selection works per block); this makes the offset look freshly produced
// there without emitting an instruction.
__device__ __forceinline__ void lane_offset_here(uint32_t& off)
{
    asm volatile("" : "+v"(off));
}

// typed accesses through such a pointer; the *_a4 forms only assume 4-byte alignment (ragged rows)
template <typename V>
__device__ __forceinline__ V gload(global_ptr<const uint8_t> p)
{
    return *(global_ptr<const V>)p;
}
template <typename V>
__device__ __forceinline__ V gload_a4(global_ptr<const uint8_t> p)
{
    typedef V __attribute__((aligned(4))) VA;
    return *(global_ptr<const VA>)p;
}
template <typename V>
__device__ __forceinline__ void gstore_nt(global_ptr<uint8_t> p, V v)
{
    __builtin_nontemporal_store(v, (global_ptr<V>)p);
}
template <typename V>
__device__ __forceinline__ void gstore_a4(global_ptr<uint8_t> p, V v)
{
    typedef V __attribute__((aligned(4))) VA;
    *(global_ptr<VA>)p = v;
}

template <typename V>
__device__ __forceinline__ void gstore_a1(global_ptr<uint8_t> p, V v)
{
    typedef V __attribute__((aligned(1))) VA;
    *(global_ptr<VA>)p = v;
}

// RAGGED rows (width % 4 != 0, or buffers that are not 16-byte aligned): a lane of an edge strip whose four pixels
// run past the row's right end loads the row's LAST four pixels instead — its load offset is clamped to pixel w - 4 —
// and moves them into place afterwards: `shift` = how far the lane starts beyond pixel w - 4.  Round 2 addressed those
// lanes' pixels one by one under a divergent branch: four more load instructions per row for the whole wave, in two
// strips of five at width 1023.  Same box, same buffers, Gaussian: 1023 x 819 x 2048 frames 4.73 -> 4.89 TB/s, 427 x 640 x
// 4096 4.90 -> 5.00, 75 x 75 x 65536 1.96 -> 2.06 (profiles/r03_ragged_ab.txt).  Needs w >= 4; narrower images keep the
// per-pixel form.  Used by gauss_slide.hip only: the same change left Sobel where it was (4.93 / 4.94 TB/s at 1023 x 819,
// 4.19 / 4.19 at 427 x 640) and cost the fused pipeline 2-3 % (3.84 against 3.91, 3.75 against 3.86), so those two keep
// the per-pixel loads.  (What does NOT matter for these rows is the alignment of the 16-byte accesses themselves: with
// every row base rounded down to 16 bytes — wrong pixels, timing only — the ragged Gaussian ran 4.77 against 4.81.)
struct RaggedEdge {
    bool sh1, sh2, sh3, sh4;  // shift >= 1, 2, 3, 4
};

// q_lane = the lane's own quad column (may be < 0 or past the row), q_load = the quad it loads (clamped by the caller).
// Returns the byte offset of the load inside a row.
__device__ __forceinline__ uint32_t ragged_edge_setup(int q_lane, int q_load, int w, RaggedEdge* e)
{
    const int xs = min(4 * q_load, w - 4);
    const int shift = 4 * q_lane - xs;  // > 0 only where the lane's pixels run past pixel w - 4
    e->sh1 = shift >= 1;
    e->sh2 = shift >= 2;
    e->sh3 = shift >= 3;
    e->sh4 = shift >= 4;
    return (uint32_t)xs * 4u;
}

// clamp-to-edge columns (Gaussian, the pipeline's gray image): out[j] = in[min(j + shift, 3)] — pixel w - 1 replicated
__device__ __forceinline__ void ragged_shift_clamp(u32x4& p, const RaggedEdge& e)
{
    p.x = e.sh3 ? p.w : (e.sh2 ? p.z : (e.sh1 ? p.y : p.x));
    p.y = e.sh2 ? p.w : (e.sh1 ? p.z : p.y);
    p.z = e.sh1 ? p.w : p.z;
}

// What every sliding kernel starts with: which (frame, band, strip) this wave owns.  Wave-uniform
// (readfirstlane keeps it in SGPRs).  Returns false for the padding waves of the last block of a phase.
struct SlideItem {
    int strip, band, y0, nout;
    uint32_t work;  // index of this work item in [0, nwork_a + nwork_b): one flag slot per item
    size_t frame;
};

__device__ __forceinline__ bool slide_item(const BandPlan& plan, int nstrips, int h, SlideItem* it)
{
    const bool tail = blockIdx.x >= plan.nblocks_a;
    const uint32_t blk = tail ? xcd_remap(blockIdx.x - plan.nblocks_a, plan.nblocks_b)
                              : xcd_remap(blockIdx.x, plan.nblocks_a);
    const uint32_t work =
        __builtin_amdgcn_readfirstlane(blk * kSlideWavesPerBlock + (uint32_t)(threadIdx.x >> 6));
    if (work >= (tail ? plan.nwork_b : plan.nwork_a))
        return false;
    const int nbands = tail ? plan.nbands_b : plan.nbands_a;
    const int band_rows = tail ? plan.rows_b : plan.rows_a;
    it->work = tail ? plan.nwork_a + work : work;
    it->strip = work % nstrips;
    const int band = (work / nstrips) % nbands;
    it->frame = work / ((uint32_t)nstrips * nbands);
    it->band = band;
    it->y0 = (tail ? plan.y_split : 0) + band * band_rows;
    it->nout = min(band_rows, (tail ? h : plan.y_split) - it->y0);
    return true;
}
#endif

}  // namespace mi355
