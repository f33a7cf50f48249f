// Shared definitions for the gfx950 rasterizer kernels: scratch-state layout,
// small device helpers, launch wrappers.  MI355X only (wave64, 256 CUs, 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define FRG_TILE 16
#define FRG_TILE_PIX 256
#define FRG_WAVE 64
#define FRG_NUM_XCD 8
#define FRG_SLOT_FLOATS 9   // per-instance backward partial: rgb(3) mean2D(2) conic(3) opacity(1)
#ifndef FRG_SLOT_STRIDE
#define FRG_SLOT_STRIDE 9   // floats from one instance's slot to the next in the backward workspace
#endif
#define FRG_REACHED_MASK 0xFF00u   // byte 1 of rgb_clamped[].w: set by the backward blend for Gaussians with a slot that may hold a gradient
// The backward blend walks a tile's processed list prefix in SEGMENTS, each segment an independent work item
// (blend_impl.h): the forward leaves every pixel's transmittance and accumulated colour at the segment boundaries it
// crosses (BinningState::ckpt, ImageState::final_C).  A multiple of 64 (the staging round), a power of two.
// Measured (round 4, same box, C3 / C4 / clustered scene, backward blend): 256 0.425 / 0.34 / 0.42 ms, 512 0.392 / 0.37 / 0.37,
// 1024 0.397 / 0.385 / 0.39, 2048 0.40 / 0.52 / 0.39; one tile per item (round 3) 0.40 / 0.58 / 0.40 -- the best length is the
// one that gives the frame about as many items as the GPU holds single-wave workgroups (16 384): a frame of 5 M instances
// (C4) wants 256, one of 16 M (C3) 512.  Round 5: the length is chosen per FRAME by the forward from the instance count
// its binning chunk is carved for (bwd_seg_log below; option "bwd_seg_log" pins it) and stamped into the image chunk's
// counters, where the backward finds it.  16 / 8 bytes of checkpoint space per instance at 256 / 512.
#define FRG_BWD_SEG_LOG_MIN 8
#define FRG_BWD_SEG_LOG_MAX 10
#define FRG_BWD_SEG_SWITCH (1u << 23)   // instances (as carved) from which the segments are 512 entries long
#define FRG_BWD_HEAVY_SLOTS (4 * 896)   // four slot windows of the per-Gaussian backward (preprocess_bwd.hip)
#define FRG_BIN_THREADS 1024     // binning workgroup = chunk of Gaussians
#define FRG_BIN_MAX_BLOCKS 256   // rows of the (workgroup x tile) count matrix: one persistent workgroup per CU
#define FRG_BIN_SEGS 8           // row segments of the column scan
// LDS bins of the preprocess (tiles + record cells), 4 bytes each, beside its static arrays -- 52 KiB of SH staging, 48 KiB
// of record assembly, 20.25 KiB of walk scratch = 123 136 B -- in the CU's 163 840 B, with 256 B to spare for whatever the
// compiler or the runtime may want (preprocess.hip asserts the sum)
#define FRG_BIN_STATIC_LDS 123136
#define FRG_BIN_MAX_LDS_TILES 10112
#define FRG_MAX_TILE_ROWS 1024       // cells (tile row x band of tile columns) of the scatter's record order; also the largest number of tile rows it handles

namespace frg {

struct Dims {
    int P, W, H, gx, gy, T;  // T = gx*gy tiles
};

__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// log2 of the backward blend's segment length for a binning chunk carved for R instances (forced: option "bwd_seg_log", 0 = by R)
__host__ __device__ inline int bwd_seg_log(size_t R, int forced = 0)
{
    if (forced >= FRG_BWD_SEG_LOG_MIN && forced <= FRG_BWD_SEG_LOG_MAX) return forced;
    return R < (size_t)FRG_BWD_SEG_SWITCH ? 8 : 9;
}

// ---- geometry chunk ---------------------------------------------------------
// One 48-byte RECORD per Gaussian = three float4, contiguous: the blend kernels gather all three per list entry,
// and as three separate arrays those were three L2 requests to three cache lines per entry (the L2 request rate of
// 16-byte gathers, not bandwidth, bounded the forward blend's staging); side by side, three records out of four sit
// inside one 128-byte line and the second and third load hit what the first one brought in.  48 bytes rather than a
// padded 64: the per-Gaussian streaming kernels write and read every byte of the array either way.
// xydr / conic_opacity / rgb_clamped point at float4 0..2 of record 0: element i of each lives at [FRG_REC * i].
#define FRG_REC 3
struct GeomState {
    float4* xydr;            // pixel x, pixel y, view depth, radius (as float, exact integer)
    float4* conic_opacity;   // conic a, b, c, opacity          (forward.cu:253)
    float4* rgb_clamped;     // r, g, b, flag word in the bit pattern of .w: bits 0-2 clamp flags (forward), byte 1 "reached" (backward blend)
    uint32_t* tiles_touched;
    // what the scatter needs of a visible Gaussian, compact (12 bytes, streamed): depth bits, tile rectangle
    // x0 | y0 << 16, x1 | y1 << 16 -- the records above are laid out for the blend kernels' gathers
    uint32_t* depth_rect;    // three planes of P words: depth bits | rect min (x | y << 16) | rect max
    uint32_t* point_offsets; // inclusive scan of tiles_touched (rasterizer_impl.cu:277)
    uint32_t* block_sums;    // per-256-Gaussian block totals -> exclusive prefix
    int* internal_radii;     // used when the caller passes radii == NULL (rasterizer_impl.cu:228-231)
    // the visible Gaussians' scatter records {depth bits, index, x0 | y0 << 16, x1 | y1 << 16}, grouped by the cell
    // (tile row, band of tile columns) of the rectangle's first tile (reorder_kernel): consecutive records touch the
    // same few dozen tiles, so the scatter's 8-byte stores into a tile's segment land within microseconds of each
    // other and leave the L2 as full lines.  In the caller's (arbitrary) order every 128-byte line of the pairs array was open for the whole
    // kernel and went to HBM as 32-byte sectors: 0.49 GB written for 0.13 GB of pairs.
    uint4* row_records;
    // d(colour)/d(view direction) of every visible Gaussian, 9 floats {ddx[3], ddy[3], ddz[3]} (ShDir, gauss_math.h):
    // written by the forward's SH pass, read by the per-Gaussian backward instead of the SH rows
    float* sh_dir;
    // [0] number of, [1 ..] wave numbers of the 64-Gaussian waves whose Gaussians own more than FRG_BWD_HEAVY_SLOTS
    // instances (near-camera Gaussians of hundreds of tiles): found where point_offsets is finished (reorder_kernel /
    // scatter_kernel), read by the per-Gaussian backward, which gives each of them a 16-wave workgroup
    uint32_t* heavy_waves;
    uint32_t* sh_layout;     // one word: 1 = the SH pass stored sh_dir by slot (sh_slot_of), 0 = by lane
    size_t bytes;
    __host__ static GeomState carve(char* base, int P)
    {
        GeomState s;
        size_t o = 0;
        size_t Pp = (size_t)(P > 0 ? P : 1);
        s.xydr = (float4*)(base + o);
        s.conic_opacity = s.xydr + 1; s.rgb_clamped = s.xydr + 2;
        o = align_up(o + Pp * 16 * FRG_REC, 256);
        s.tiles_touched = (uint32_t*)(base + o); o = align_up(o + Pp * 4, 256);
        s.depth_rect = (uint32_t*)(base + o); o = align_up(o + Pp * 12, 256);
        s.point_offsets = (uint32_t*)(base + o); o = align_up(o + Pp * 4, 256);
        s.block_sums = (uint32_t*)(base + o); o = align_up(o + ((Pp + 255) / 256 + 1) * 4, 256);  // per chunk
        s.internal_radii = (int*)(base + o); o = align_up(o + Pp * 4, 256);
        s.row_records = (uint4*)(base + o); o = align_up(o + Pp * 16, 256);
        s.sh_dir = (float*)(base + o); o = align_up(o + ((Pp + 63) / 64 * 64) * 36, 256);   // whole waves: the SH pass stores 16-Gaussian blocks
        s.heavy_waves = (uint32_t*)(base + o); o = align_up(o + (Pp / 64 + 2) * 4, 256);
        s.sh_layout = (uint32_t*)(base + o); o = align_up(o + 4, 256);
        s.bytes = o;
        return s;
    }
};

// Rows of GeomState::sh_dir inside a wave's block of 64: by lane -- or, when the forward ran the SH pass that streams
// only the visible Gaussians (GeomState::sh_layout = 1: views that see a part of the model) and less than three quarters
// of the wave are visible, by rank among the visible ones.  The forward's SH pass and the per-Gaussian backward both
// derive the slot from the wave's visibility ballot.
__host__ __device__ inline bool sh_slot_dense(int nvis) { return nvis > 48; }   // four sub-batches of 16 either way
__device__ __forceinline__ int sh_slot_of(uint64_t vis, int lane, bool sparse_layout)
{
    return (!sparse_layout || sh_slot_dense(__popcll(vis))) ? lane : __popcll(vis & ((1ull << lane) - 1ull));
}

// ---- image chunk ----------------------------------------------------------
#define FRG_BWD_LEN_BUCKETS 32   // length buckets of the tiles' last segments (backward blend items, longest first)
#define FRG_SORT_CLASSES 5    // tile-list size classes of the sort: <=512, <=2048, <=4096, <=8192, >8192
struct Counters {            // written by the forward's kernels; 56 bytes, posted to / read back by the host
    uint32_t num_rendered;
    uint32_t max_tile_count;
    uint32_t filtered;       // prefiltered assertion (auxiliary.h:154-162)
    uint32_t overflow;       // deferred-counters forward: num_rendered exceeded the caller's capacity
    uint32_t class_count[FRG_SORT_CLASSES];  // number of tiles per sort size class
    // modes of the forward that filled this image chunk, stamped by scan_kernel: the backward follows
    // THESE, not the process-wide options at the time it is called
    uint32_t tight_binning;
    uint32_t num_visible;    // Gaussians with at least one tile (records in GeomState::row_records)
    // the instance count the forward CARVED its binning chunk with (num_rendered; the capacity of a deferred-counters
    // forward), stamped by the chunk scan: BinningState::ckpt sits behind point_list and pairs, at an offset that depends
    // on it, and the backward blend takes the offset from HERE -- not from the R its caller passes, which may be either
    // of the two (a wrong offset would silently read other tiles' checkpoints for every walk deeper than one segment)
    uint32_t carved_R;
    uint32_t bwd_seg_log;    // log2 of the segment length the forward blend left its checkpoints at (stamped by blend_fwd_kernel)
    // the forward's blend modes, stamped by blend_fwd_kernel: FRG_FWD_STAMPED | FRG_FWD_EXACT (the reference's arithmetic) |
    // FRG_FWD_ONLY (frg_forward_args::forward_only: nothing was kept for a backward).  A backward whose host side does
    // not know the forward that filled its buffers (a geometry buffer cloned or restored at another address, a process that
    // forgot it) reads THIS word back (api.hip) instead of falling back to a process option
    uint32_t fwd_flags;
};
#define FRG_FWD_EXACT 1u
#define FRG_FWD_ONLY 2u
#define FRG_FWD_STAMPED 0x80000000u
static_assert(sizeof(Counters) == 56, "Counters: 14 words (the mailbox's second line is 8 + 56 bytes)");
// Pinned HOST memory the scan workgroups write with system-scope stores, polled by the forward's host thread: the
// instance count as soon as the chunk scan has it (the host sizes the binning buffer and enqueues the scatter while the
// reorder is still running), the full counters when the tile scan is done (the sort's grid sizes).  Replaces the
// copy kernel + stream synchronisation of the read-back (~28 us of idle GPU per forward, a fifth of C2's step).
// seq_*: the forward's sequence number, stored last.
struct Mailbox {
    uint32_t seq_r, num_rendered, pad0[14];      // one 64-byte line per stage
    uint32_t seq_c, pad1[1];
    Counters c;                                  // (8 + 56 bytes: the second line)
#define FRG_MAILBOX_HEAVY_OFFSET 128
    // third post, by the scatter (not waited for): how many 64-Gaussian waves own more than FRG_BWD_HEAVY_SLOTS
    // backward slots -- a backward that finds its forward's post here and reads 0 skips the 16-wave launch of the
    // per-Gaussian backward and its fork / join (~11 us per step at C3)
    // ONE 8-byte word, stored at once -- (sequence number << 32) | heavy waves: the scatters of two forwards that a
    // thread put on different streams may post in any order, and a torn pair would hand one forward's count to the other
    unsigned long long heavy_post;
    uint32_t visible, pad2[13];                 // visible: Counters::num_visible (the next forward's sparse_sh hint only)
};
static_assert(offsetof(Mailbox, c) == 72 && offsetof(Mailbox, heavy_post) == FRG_MAILBOX_HEAVY_OFFSET && sizeof(Mailbox) == 192,
              "Mailbox: one 64-byte line per post");
__device__ __forceinline__ void mailbox_post(uint32_t* flag, uint32_t seq)
{
    __threadfence_system();
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__host__ __device__ inline int sort_class_of(uint32_t n)
{
    return n <= 512 ? 0 : n <= 2048 ? 1 : n <= 4096 ? 2 : n <= 8192 ? 3 : 4;
}

// class * 8 + bucket, bucket 0 = the longest eighth of the class's size range
__host__ __device__ inline int sort_subclass_of(uint32_t n)
{
    const int cls = sort_class_of(n);
    const uint32_t lo = cls == 0 ? 0u : cls == 1 ? 512u : cls == 2 ? 2048u : cls == 3 ? 4096u : 8192u;
    const uint32_t span = cls == 0 ? 512u : cls == 1 ? 1536u : cls == 2 ? 2048u : cls == 3 ? 4096u : 65536u;
    uint32_t k = (n - lo - 1u) * 8u / span;
    if (k > 7u) k = 7u;
    return cls * 8 + (7 - (int)k);
}

struct ImageState {
    float* final_T;          // accum_alpha in the reference (rasterizer_impl.h:47)
    uint32_t* n_contrib;
    uint2* ranges;           // per tile [start,end) into point_list; (0,0) when empty
    uint32_t* tile_count;    // instances per tile: written by the tile scan (LDS bins) | cleared by the forward's memset and counted with atomics (global bins)
    uint32_t* tile_fill;     // scatter cursor; zero at the start of every forward (colsum_kernel, or the memset of the global-bins path)
    uint32_t* tile_work;     // list entries the forward blend walked (max over the tile's pixels): stored by the tile's last wave
    // Work items of the backward blend, listed by the FORWARD blend's tile workgroups as they finish (round 5: there is no
    // ordering kernel in front of the backward any more): per XCD band of tiles (frg_common.h: xcd_of_tile)
    //   bwd_cnt  [8][FRG_BWD_LEN_BUCKETS + 1]  tiles per length bucket of their LAST segment (bucket 0 = the longest), and in
    //            [x][FRG_BWD_LEN_BUCKETS] the number of full-segment items of the band (BinningState::bwd_full);
    //            zero at the start of every forward (colsum_kernel / the memset of the global-bins path)
    //   bwd_last [8][FRG_BWD_LEN_BUCKETS][bwd_cap_b]  the tiles of each bucket, in the order their workgroups finished
    uint32_t* bwd_cnt;
    uint32_t* bwd_last;
    uint32_t bwd_cap_b;      // tiles per XCD band, at most
    // accumulated colour (without the background term) of every pixel of a tile whose walk crossed a segment boundary:
    // float4[T * 256], quadrant-major like BinningState::ckpt.  The backward blend starts a segment that is not a
    // pixel's last from S = dL/dC . (final_C - colour accumulated at the segment's end) + T_final (bg . dL/dC)
    float4* final_C;
    Counters* counters;      // every field written by the forward's kernels (filtered: set-only, cleared when the assertion is on)
    uint2* cutoff;           // per tile: (depth bits, index) of the last instance the backward blend processed
    uint32_t* bin_matrix;    // [FRG_BIN_MAX_BLOCKS][T] per-workgroup tile counts (-> scatter bases when !row_order)
    uint32_t* seg_sums;      // [FRG_BIN_SEGS][T]
    uint32_t* row_matrix;    // [FRG_BIN_MAX_BLOCKS][ncells] per-workgroup counts of visible Gaussians by the cell of their rectangle's first tile -> bases of reorder_kernel
    uint32_t* row_start;     // [ncells] visible Gaussians per cell (reorder_kernel scans them)
    bool lds_bins;           // false: image too large for LDS histograms -> global-atomic binning
    bool row_order;          // scatter over cell-ordered records (needs lds_bins)
    // cells of the record order: tile row x band of band_w tile columns (nbands per row, at most FRG_MAX_TILE_ROWS cells)
    int band_w, nbands, ncells;
    uint32_t* class_tiles;   // [FRG_SORT_CLASSES + 1][T] tile ids per sort size class (longest first inside a class); last row: the empty tiles
    size_t zero_begin, zero_bytes;  // region [tile_count .. counters]: one memset in the global-bins path (api.hip)
    size_t bytes;
    __host__ static ImageState carve(char* base, int W, int H, bool force_global_bins = false)
    {
        ImageState s;
        size_t N = (size_t)W * H;
        size_t T = (size_t)((W + FRG_TILE - 1) / FRG_TILE) * ((H + FRG_TILE - 1) / FRG_TILE);
        size_t o = 0;
        s.final_T = (float*)(base + o); o = align_up(o + N * 4, 256);
        s.n_contrib = (uint32_t*)(base + o); o = align_up(o + N * 4, 256);
        s.ranges = (uint2*)(base + o); o = align_up(o + T * 8, 256);
        s.cutoff = (uint2*)(base + o); o = align_up(o + T * 8, 256);
        s.zero_begin = o;
        s.tile_count = (uint32_t*)(base + o); o = align_up(o + T * 4, 256);
        s.tile_fill = (uint32_t*)(base + o); o = align_up(o + T * 4, 256);
        s.tile_work = (uint32_t*)(base + o); o = align_up(o + T * 4, 256);
        s.bwd_cnt = (uint32_t*)(base + o); o = align_up(o + (size_t)FRG_NUM_XCD * (FRG_BWD_LEN_BUCKETS + 1) * 4, 256);
        s.counters = (Counters*)(base + o); o = align_up(o + sizeof(Counters), 256);
        s.zero_bytes = o - s.zero_begin;
        s.bwd_cap_b = (uint32_t)(T / FRG_NUM_XCD + 1);
        s.bwd_last = (uint32_t*)(base + o); o = align_up(o + (size_t)FRG_NUM_XCD * FRG_BWD_LEN_BUCKETS * s.bwd_cap_b * 4, 256);
        s.final_C = (float4*)(base + o); o = align_up(o + T * FRG_TILE_PIX * 16, 256);
        s.class_tiles = (uint32_t*)(base + o); o = align_up(o + (size_t)(FRG_SORT_CLASSES + 1) * T * 4, 256);
        const size_t gy = (size_t)((H + FRG_TILE - 1) / FRG_TILE);
        s.lds_bins = T <= FRG_BIN_MAX_LDS_TILES && !force_global_bins;
        s.bin_matrix = nullptr; s.seg_sums = nullptr; s.row_matrix = nullptr; s.row_start = nullptr;
        if (s.lds_bins) {
            s.seg_sums = (uint32_t*)(base + o); o = align_up(o + (size_t)FRG_BIN_SEGS * T * 4, 256);
            s.bin_matrix = (uint32_t*)(base + o); o = align_up(o + (size_t)FRG_BIN_MAX_BLOCKS * T * 4, 256);
        }
        const size_t gx = (size_t)((W + FRG_TILE - 1) / FRG_TILE);
        s.nbands = (int)(gy ? FRG_MAX_TILE_ROWS / gy : 1);
        if (s.nbands < 1) s.nbands = 1;
        if ((size_t)s.nbands > gx) s.nbands = (int)gx;
        s.band_w = (int)((gx + s.nbands - 1) / s.nbands);
        s.nbands = (int)((gx + s.band_w - 1) / s.band_w);
        s.ncells = (int)gy * s.nbands;
        s.row_order = s.lds_bins && gy <= FRG_MAX_TILE_ROWS && T + (size_t)s.ncells <= FRG_BIN_MAX_LDS_TILES;
        if (s.row_order) {
            s.row_matrix = (uint32_t*)(base + o); o = align_up(o + (size_t)FRG_BIN_MAX_BLOCKS * s.ncells * 4, 256);
            s.row_start = (uint32_t*)(base + o); o = align_up(o + ((size_t)s.ncells + 1) * 4, 256);
        }
        s.bytes = o;
        return s;
    }
};

// ---- binning chunk --------------------------------------------------------
#define FRG_SORT_LDS_CAP 8192   // largest tile list sorted entirely in LDS
// lists of FRG_SORT_LDS_CAP + 1 .. FRG_SORT_MID_MAX entries: sorted chunks of FRG_SORT_CHUNK entries, every
// FRG_SORT_SAMPLE-th entry a sample, splitters from the sorted samples (sort.hip); longer lists: global LSD passes
#define FRG_SORT_CHUNK FRG_SORT_LDS_CAP
#define FRG_SORT_SAMPLE 64
#define FRG_SORT_MID_MAX (FRG_SORT_CHUNK * (FRG_SORT_CHUNK / FRG_SORT_SAMPLE / 2))   // 64 chunks: 524288 entries, 8192 samples

// Work lists and tables of the splitter sort (uint32 words at BinningState::big_plan):
//   hdr[0] chunk work items  hdr[1] buckets queued so far  hdr[2] lists planned  hdr[3] diagnostics
//   chunks   uint2 {list, chunk}              one per chunk of every long list: at most R / CHUNK + #lists
//   buckets  uint4 {list, bucket, first output position inside the list, entries}: a list of m chunks has at most
//            128 m / (128 - m) + 1 buckets: fewer than n / 4064 + 1
//   lists    uint4 {tile, table offset, m, buckets}
//   tables   pos[bucket][chunk] = entries of the chunk at or below the bucket's upper splitter: buckets * m <= 129 m words
struct BigPlan {
    uint32_t* hdr; uint2* chunks; uint4* buckets; uint4* lists; uint32_t* tables;
    __host__ __device__ static size_t max_lists(size_t R) { return R / FRG_SORT_CHUNK + 1; }
    __host__ __device__ static size_t words(size_t R)
    {
        const size_t nl = max_lists(R);
        return 64 + 2 * (2 * nl + 2) + 4 * (4 * nl + 4) + 4 * (nl + 1) + 129 * 2 * nl + 64;
    }
    __host__ __device__ static BigPlan carve(uint32_t* base, size_t R)
    {
        BigPlan p;
        const size_t nl = max_lists(R);
        p.hdr = base;
        p.chunks = reinterpret_cast<uint2*>(base + 64);
        p.buckets = reinterpret_cast<uint4*>(base + 64 + 2 * (2 * nl + 2));
        p.lists = p.buckets + (4 * nl + 4);
        p.tables = reinterpret_cast<uint32_t*>(p.lists + (nl + 1));
        return p;
    }
};

struct BinningState {
    uint32_t* point_list;    // sorted Gaussian indices, tile-major
    uint2* pairs;            // (depth bits, index), tile-major, scatter order
    // Forward-blend checkpoints for the segmented backward blend: {T, C0, C1, C2} of every pixel of a tile at the list
    // positions k * SEG (k >= 1; SEG = 1 << seg_log) its walk passes -- the state BEFORE entry k * SEG is blended.  Record
    // (first / SEG + k) belongs to the tile whose list starts at instance `first`: unique, because a list of n
    // entries has floor((n - 1) / SEG) boundaries and the next list starts n instances later.  256 float4 per record,
    // quadrant-major (the forward's wave q writes [q * 64, q * 64 + 64)).  16 / 8 bytes of space per instance at 256 / 512; written only where crossed.
    float4* ckpt;
    __host__ __device__ static size_t ckpt_records(size_t R, int seg_log) { return (R >> seg_log) + 2; }
    // The backward blend's FULL-segment items (tile, segment), listed per XCD band by the forward blend as its tiles finish:
    // [8][full_cap(R)] -- a band may hold every long list of the frame, and all lists together have at most R / segment
    // boundaries.  Sized for the shortest segment length.  R / 4 bytes.
    uint2* bwd_full;
    __host__ __device__ static size_t full_cap(size_t R) { return (R >> FRG_BWD_SEG_LOG_MIN) + 2; }
    // byte offsets inside a chunk carved for R instances -- the one place that knows them: carve() and the backward blend,
    // which evaluates them on the device from Counters::carved_R (the R a backward is CALLED with may be the frame's
    // instance count or a deferred forward's capacity)
    __host__ __device__ static size_t full_offset(size_t R)
    {
        const size_t Rr = R > 0 ? R : 1;
        return align_up(Rr * 4, 256) + align_up(Rr * 8, 256);
    }
    __host__ __device__ static size_t ckpt_offset(size_t R)
    {
        const size_t Rr = R > 0 ? R : 1;
        return full_offset(Rr) + align_up((size_t)FRG_NUM_XCD * full_cap(Rr) * 8, 256);
    }
    uint2* pairs_tmp;        // second pair buffer (sorted chunks / ping-pong), only when some tile exceeds the LDS capacity
    uint32_t* big_hist;      // digit counters of the LSD sort of lists beyond FRG_SORT_MID_MAX: one 256-entry row per 1024 elements
    uint32_t* big_plan;      // BigPlan of the splitter sort
    size_t bytes;
    // rows of big_hist: row r of the tile whose list starts at element x has the id (x >> 10) + (x >> 13) + r --
    // unique and increasing, because only lists longer than 8192 = 2^13 entries own rows
    __host__ __device__ static size_t big_hist_rows(size_t R) { return (R >> 10) + (R >> 13) + 2; }
    __host__ __device__ static size_t big_hist_row(uint32_t first, uint32_t r) { return (size_t)(first >> 10) + (first >> 13) + r; }
    int seg_log;             // log2 of the segment length ckpt is sized for
    size_t carved_R;         // the instance count this chunk was carved for (>= 1)
    __host__ static BinningState carve(char* base, int R, int max_tile_count, int forced_seg_log = 0)
    {
        BinningState s;
        size_t Rr = (size_t)(R > 0 ? R : 1);
        size_t o = 0;
        s.seg_log = bwd_seg_log(Rr, forced_seg_log);
        s.carved_R = Rr;
        s.point_list = (uint32_t*)(base + o); o = align_up(o + Rr * 4, 256);
        s.pairs = (uint2*)(base + o); o = align_up(o + Rr * 8, 256);
        s.bwd_full = (uint2*)(base + full_offset(Rr));
        s.ckpt = (float4*)(base + ckpt_offset(Rr));
        o = ckpt_offset(Rr) + align_up(ckpt_records(Rr, s.seg_log) * FRG_TILE_PIX * 16, 256);
        s.pairs_tmp = nullptr;
        s.big_hist = nullptr;
        s.big_plan = nullptr;
        if (max_tile_count > FRG_SORT_LDS_CAP) {
            s.pairs_tmp = (uint2*)(base + o); o = align_up(o + Rr * 8, 256);
            s.big_plan = (uint32_t*)(base + o); o = align_up(o + BigPlan::words(Rr) * 4, 256);
            // (the deferred-counters forward passes FRG_SORT_LDS_CAP + 1: "unknown" -- sized for every path)
            if (max_tile_count > FRG_SORT_MID_MAX || max_tile_count == FRG_SORT_LDS_CAP + 1) {
                s.big_hist = (uint32_t*)(base + o); o = align_up(o + big_hist_rows(Rr) * 256 * 4, 256);
            }
        }
        s.bytes = o;
        return s;
    }
};

// ---- per-view constants ------------------------------------------------------
// Scalars travel by value; the matrices stay where the caller put them (device
// memory, as in the reference) and are read through wave-uniform scalar loads.
struct ViewParams {
    float tan_fovx, tan_fovy, focal_x, focal_y, scale_modifier;
    int W, H, gx, gy;
    int D, M;   // active SH degree, coefficients per channel in memory
    int tight;  // 1: binning keeps only (Gaussian, tile) instances that can reach alpha >= 1/255 in the tile
    int sparse_sh;  // 1: the view is expected to see a part of the model only (occlusion mask, last view's count): SH pass over the visible Gaussians
    int sh_no_dir;  // 1: the SH pass leaves d(colour)/d(direction) out (option sh_dir_in_backward; a forward_only call): GeomState::sh_layout bit 1
};
struct ViewMats {
    float view[16];
    float proj[16];
    float campos[3];
};
__device__ __forceinline__ void load_view_mats(const float* __restrict__ view, const float* __restrict__ proj,
                                               const float* __restrict__ campos, ViewMats& m)
{
#pragma unroll
    for (int i = 0; i < 16; i++) { m.view[i] = view[i]; m.proj[i] = proj[i]; }
    m.campos[0] = campos[0]; m.campos[1] = campos[1]; m.campos[2] = campos[2];
}

// XCD-aware tile mapping: workgroup b runs on XCD b % 8 (observed dispatch
// order); give every XCD a contiguous band of tile rows so that neighbouring
// tiles -- which share most of their Gaussians -- hit the same 4 MiB L2.
__device__ __forceinline__ int xcd_tile_of_block(int b, int T)
{
    const int per = T / FRG_NUM_XCD, rem = T % FRG_NUM_XCD;
    const int xcd = b % FRG_NUM_XCD, k = b / FRG_NUM_XCD;
    // XCDs [0,rem) own per+1 tiles, the rest own per tiles.
    const int start = xcd * per + (xcd < rem ? xcd : rem);
    const int mine = per + (xcd < rem ? 1 : 0);
    if (k < mine) return start + k;
    return -1;  // padding block of the rounded-up grid
}
// inverse of xcd_tile_of_block: the XCD whose band holds tile t
__device__ __forceinline__ int xcd_of_tile(int t, int T)
{
    const int per = T / FRG_NUM_XCD, rem = T % FRG_NUM_XCD;
    const int cut = rem * (per + 1);
    return t < cut ? t / (per + 1) : rem + (t - cut) / (per > 0 ? per : 1);
}
__host__ inline int xcd_grid_blocks(int T) { return ((T + FRG_NUM_XCD - 1) / FRG_NUM_XCD) * FRG_NUM_XCD; }

// k / w and k % w for 0 <= k < 2^20, 1 <= w < 2^10 (positions inside a tile rectangle) without the ~25-instruction
// integer division: (k + 0.5) * rcp(w) is within 1e-6 relative of (k + 0.5) / w, whose distance to the next integer
// is at least 0.5 / w >= 5e-4 -- the truncation is exact.
__device__ __forceinline__ void rect_divmod(uint32_t k, uint32_t w, uint32_t& q, uint32_t& r)
{
    q = (uint32_t)(((float)k + 0.5f) * __builtin_amdgcn_rcpf((float)w));
    r = k - q * w;
}

// wave64 inclusive prefix sum on DPP moves (row_shr 1/2/4/8 inside the 16-lane rows, then row_bcast:15 and
// row_bcast:31 across them): ~7 cycles per step against ~24 for a ds_bpermute shuffle (tools/micro/pk_rate.hip)
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v)
{
#define FRG_SCAN_STEP(CTRL, ROWMASK) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, false);
    FRG_SCAN_STEP(0x111, 0xf) FRG_SCAN_STEP(0x112, 0xf) FRG_SCAN_STEP(0x114, 0xf) FRG_SCAN_STEP(0x118, 0xf)
    FRG_SCAN_STEP(0x142, 0xa) FRG_SCAN_STEP(0x143, 0xc)
#undef FRG_SCAN_STEP
    return v;
}

// float -> int exactly like the reference's C cast on the GPU (v_cvt_i32_f32).
__device__ __forceinline__ int f2i(float v) { return (int)v; }

__device__ __forceinline__ void tile_rect(float px, float py, int max_radius, int gx, int gy, int& x0, int& y0, int& x1, int& y1)
{
    // auxiliary.h:46-56 getRect, same float expressions and truncating casts
    x0 = min(gx, max(0, f2i((px - max_radius) / FRG_TILE)));
    y0 = min(gy, max(0, f2i((py - max_radius) / FRG_TILE)));
    x1 = min(gx, max(0, f2i((px + max_radius + FRG_TILE - 1) / FRG_TILE)));
    y1 = min(gy, max(0, f2i((py + max_radius + FRG_TILE - 1) / FRG_TILE)));
}

// ---- exact quadrant culling ---------------------------------------------------------
// Can any pixel of the 8x8 quadrant whose first pixel is (qx0, qy0) reach alpha >= 1/255
// for the Gaussian (centre x,y; conic a,b,c; opacity o)?
// alpha = min(0.99, o * exp(power)), power = -1/2 Q(d), Q(d) = a dx^2 + 2 b dx dy + c dy^2, so
// alpha < 1/255 on the whole quadrant whenever 1/2 min_rect Q > ln(255 o), the minimum taken
// over the continuous rectangle spanned by the quadrant's pixel centres (a superset of the
// pixels, hence conservative).  Q is convex: if the centre lies inside the rectangle the
// minimum is 0, otherwise it sits on an edge facing the centre, where Q restricted to the edge is
// a 1-D parabola whose clamped vertex gives the edge minimum in closed form.  Margins (0.1 %
// relative, 0.02 absolute on a threshold <= 5.6) dominate the rounding of the per-pixel
// evaluation (|error| <= ~1e-6 * lambda_max * d^2, lambda_max <= 1/0.3 by the low-pass,
// d^2 <= 512), so the cull never removes a pixel the reference would have blended.
// Evaluated WITHOUT contraction so that the blend kernels and the per-Gaussian backward
// (which re-derives which quadrant slots exist) agree bit for bit.
template <int EXTENT_X, int EXTENT_Y>
__device__ __forceinline__ bool rect_hit_xy(float x, float y, float4 co, int qx0, int qy0)
{
#pragma clang fp contract(off)
    const float o = co.w;
    if (!(o >= 1.0f / 255.0f)) return o != o;  // exp(power) <= 1 => alpha <= o < 1/255 everywhere
    const float a = co.x, b = co.y, c = co.z;
    // not a proper positive-definite conic (or not finite): no bound, keep
    if (!(a > 0.f) || !(c > 0.f) || !(a * c - b * b > 0.f) || !(a < 3.0e38f) || !(c < 3.0e38f)) return true;
    // 255 o lies in [1, 255]: the raw v_log_f32 (log2, 1 ulp) needs none of the denormal scaling __logf carries
    const float thr = 0.6931471805599453f * __builtin_amdgcn_logf(255.0f * o) + 0.02f;
    const float xlo = (float)qx0 - x, ylo = (float)qy0 - y, xhi = xlo + (float)EXTENT_X, yhi = ylo + (float)EXTENT_Y;
    if (xlo <= 0.f && xhi >= 0.f && ylo <= 0.f && yhi >= 0.f) return true;  // centre inside: Q = 0
    // v_rcp_f32 (1 ulp) instead of two IEEE divisions (~11 instructions each): an error of the clamped vertex
    // position enters Q at second order, far below the margins
    const float ia = __builtin_amdgcn_rcpf(a), ic = __builtin_amdgcn_rcpf(c);
    // The minimiser lies on the line dx = ex or on the line dy = ey, ex / ey the rectangle's x / y closest to the
    // centre (0 clamped into the range): from any other point of the rectangle a small step towards the centre stays
    // inside and lowers Q.  Two clamped 1-D parabola vertices instead of the four edges.
    const float ex = fminf(fmaxf(0.f, xlo), xhi), ey = fminf(fmaxf(0.f, ylo), yhi);
    const float dy = fminf(fmaxf(-b * ex * ic, ylo), yhi);
    const float dx = fminf(fmaxf(-b * ey * ia, xlo), xhi);
    const float best = fminf(a * ex * ex + 2.f * b * ex * dy + c * dy * dy, a * dx * dx + 2.f * b * dx * ey + c * ey * ey);
    return !(0.5f * 0.999f * best > thr);
}
template <int EXTENT>
__device__ __forceinline__ bool rect_hit(float x, float y, float4 co, int qx0, int qy0) { return rect_hit_xy<EXTENT, EXTENT>(x, y, co, qx0, qy0); }
__device__ __forceinline__ bool quadrant_hit(float x, float y, float4 co, int qx0, int qy0) { return rect_hit<7>(x, y, co, qx0, qy0); }
// the same bound over a whole 16x16 tile (a superset of its four quadrants)
__device__ __forceinline__ bool tile_hit(float x, float y, float4 co, int tx, int ty) { return rect_hit<FRG_TILE - 1>(x, y, co, tx * FRG_TILE, ty * FRG_TILE); }

}  // namespace frg
