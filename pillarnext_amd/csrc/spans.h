// spans.h -- shared declarations of the one-pass grouping front end (chunk_sort.hip) and its consumer (pfn_spans.hip).
//
// Vocabulary (reference semantics: pillar_encoder.py:95-125 -- the voxel index of every point, torch.unique and its inverse):
//   cell    index of a pillar position in the memory order of the dense NHWC canvas: (b * gy + yi) * gx + xi
//   slab    kSlabCells consecutive cells of one frame (the last slab of a frame may be cut); slabs tile the canvas in address order
//   chunk   kChunk consecutive rows of the point buffer, sorted by slab INSIDE LDS by one workgroup of k_chunk_sort and written
//           back as one contiguous piece of 32-byte records -- the points are read ONCE, nothing is scattered through HBM
//   row     one line of the run table: for one (chunk, frame) pair the exclusive prefix of the chunk's points over the frame's
//           slabs (u16), so that "the records of chunk c inside slabs [fa, fb)" is the contiguous run [row[fa], row[fb]) of the
//           chunk's piece.  A chunk whose points all belong to one frame (a collated batch, collate.py:15-22, is sorted by
//           sample) owns exactly one row, row c; further frames of a chunk take rows from an overflow pool
//   span    up to kSpanMaxSlabs consecutive slabs of one frame that k_span_pfn (pfn_spans.hip) groups, ranks and runs the PFN on
//           inside LDS.  The carve aims at what the LDS record slots hold (~quota points); a span with more than kSpanPillars
//           pillars (possible when nearly every point is a pillar of its own) is consumed in slices of kSpanPillars pillar ranks
#pragma once
#include "pnx_common.h"

constexpr int kSlabShift = 9;
constexpr int kSlabCells = 1 << kSlabShift;   // 512 cells = 64 KiB of a 16-bit canvas
constexpr int kSpanMaxSlabs = 16;             // 8192 cells: the span's occupancy bitmap is 256 words of LDS
constexpr int kSpanPillars = 512;             // pillars one pass over a span handles (LDS arrays)
constexpr int kChunk = 2048;                  // points per chunk: 64 KiB of records in LDS (1536 = three workgroups per CU: same reader time)
constexpr int kSpanQuota = 640;               // a span ends where the running total of points crosses a multiple of the quota ...
constexpr int kSpanSolo = 0;                  // ... and (if > 0) a slab with more points than this is a span of its own

struct SpanGeom {
  int nf;       // slabs per frame = ceil(gx * gy / kSlabCells)
  int cpf;      // cells per frame = gx * gy
  int tabw;     // u16 entries per table row: nf + 1 (entry nf = points of the row), rounded up to an even number
  int nchunks;  // ceil(n_points / kChunk): rows [0, nchunks) belong to the chunks, rows [nchunks, nchunks + ovf_cap) are the pool
  int ovf_cap;
  int B;
  int quota, solo;  // carve rule (kSpanQuota / kSpanSolo; PNX_SPAN_QUOTA / PNX_SPAN_SOLO for experiments)
};

// counters (int32) used by this path, inside the reader's counter block: [0] P (pillars; canvas-only calls add them up span by span),
// [1] N' (kept points), [3] pillars of > 32 points listed, [4] pillars outside the fp16x3 range listed, [6] slots of the 64-byte spill
// stream handed out, [7] overflow rows handed out
enum { kCntP = 0, kCntKept = 1, kCntBig = 3, kCntOvf16 = 4, kCntSpill = 6, kCntRows = 7 };

struct SpanTables {
  const uint4* recs;            // chunk-sorted records [x y z f3 | f4 f5 point-index cell]
  const uint16_t* tab;          // run table, tabw entries per row
  const int32_t* rowframe;      // frame of the row (-1: unused)
  const uint32_t* rowbase;      // first record of the row's run block
  const int32_t *frame_lo, *frame_hi;  // per frame: nchunks - (first chunk whose own row is this frame), (last such chunk) + 1; 0 = none
  const uint2* span_desc;       // per frame nf + 1 entries: {first slab of span j, points of the frame in front of it}, then {nf, points of the frame}
  const int32_t* nspan;         // spans per frame
};
