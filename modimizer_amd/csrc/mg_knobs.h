/* mg_knobs.h — every environment knob of the library, read ONCE (pthread_once) into one struct.
 *
 * None is needed in production.  Test knobs force code paths through the same parity checks (tests/, tools/test_paths.sh);
 * development knobs serve the sweeps under tools/.  A field holds the variable's value as parsed by atol (), or
 * MG_KNOB_UNSET when the variable is absent; what an absent or out-of-range value means is decided where the knob is used,
 * next to the default it replaces.  mgReloadKnobs () reads the environment again -- for tests that set a knob between calls
 * of one process; not to be called while another thread is inside the library. */
#ifndef MG_KNOBS_H
#define MG_KNOBS_H
#include <limits.h>
#ifdef __cplusplus
extern "C" {
#endif
#define MG_KNOB_UNSET LONG_MIN
typedef struct {
  /* test knobs */
  long findPath;           /* MODGPU_FIND_PATH: 'p' partitioned lookups (one level), '2' two levels, 'd' direct probes (the first letter), else automatic */
  long tablePath;          /* MODGPU_TABLE_PATH: 'd' direct atomics, 'b' bucketed (the first letter), else automatic */
  long partPacked;         /* MODGPU_PART_PACKED: 0 = wide partition elements */
  long partBig;            /* MODGPU_PART_BIG: 0 = sub-chunks of MG_PART_SUB everywhere */
  long addChunk;           /* MODGPU_ADD_CHUNK: modimizers per insert pass */
  long scanGrid;           /* MODGPU_SCAN_GRID: workers per scan launch */
  long scanGeneric;        /* MODGPU_SCAN_GENERIC: 1 = no filter mode */
  long minTile;            /* MODGPU_MIN_TILE: dev, positions per tile of the minimizer scan (default 512) */
  long minTiled;           /* MODGPU_MIN_TILED: 0 = the minimizer scan walks every window one link at a time (the path of windows wider than 256) */
  long scanDiv64;          /* MODGPU_SCAN_DIV64: 1 = the exact modes test divisibility in 64-bit arithmetic whatever k and d */
  long scanHist;           /* MODGPU_SCAN_HIST: 0 = the compaction kernel counts the first partition digit */
  long noSegmentInput;     /* MODGPU_NO_SEGMENT_INPUT: 1 = the build always gets a dense copy */
  long rankSliceShift;     /* MODGPU_RANK_SLICE_SHIFT */
  long flagPolarity;       /* MODGPU_FLAG_POLARITY: 0 / 1 force which way round the first-occurrence flags are written */
  long mergeSlots;         /* MODGPU_MERGE_SLOTS: 0 / 1: the merge kernel probes / takes the dedup kernel's slots */
  long bucketR, bucketT;   /* MODGPU_BUCKET_R / MODGPU_BUCKET_T: slots per bucket / threads of the bucket kernels */
  long hotSplit, hotChunk; /* MODGPU_HOT_SPLIT="split[,chunk]": occurrences above which a bucket is reduced chunk-wise first */
  long noAvx2;             /* MODGPU_NO_AVX2: 1 = portable packer / parser loops */
  long textHost;           /* MODGPU_TEXT_HOST: 1 = the host parser for every file */
  long textWindowKb;       /* MODGPU_TEXT_WINDOW_KB: text per window of the device parser */
  long fileBatchMbp;       /* MODGPU_FILE_BATCH_MBP */
  long fileBatchBases;     /* MODGPU_FILE_BATCH_BASES: bases per batch of the file entry points */
  long queryHostChain;     /* MODGPU_QUERY_HOST_CHAIN: 1 = modmap's chaining on the host */
  long iterHostBelow;      /* MODGPU_ITER_HOST_BELOW: modRCiterator's crossover in bases */
  /* development knobs */
  long scatterGrid;        /* MODGPU_SCATTER_GRID */
  long segSlack;           /* MODGPU_SEG_SLACK: the scan's segment room as a multiple of the fair share (1..8, default 3) */
  long partDigits;         /* MODGPU_PART_DIGITS: 0 = the second partition pass counts its digits from the elements (8 bytes each) instead of from the digit bytes the first pass leaves beside them */
  long findBits;           /* MODGPU_FIND_BITS: bits of the partitioned lookup's digit (3..9) */
  long tableLoad;          /* MODGPU_TABLE_LOAD: per cent */
  long find8;              /* MODGPU_FIND8: 0 = the two-level lookups stream the 16-byte table itself, not its 8-byte copy */
  long tightLoad;          /* MODGPU_TIGHT_LOAD: per cent; the load a set built by one add into an empty table is brought to after the dedup kernel
                              has counted its entries (0: off, the table keeps the size its occurrences' bound gave it) */
  long packThreads;        /* MODGPU_PACK_THREADS */
  long parseThreads;       /* MODGPU_PARSE_THREADS */
  long gzipThreads;        /* MODGPU_GZIP_THREADS: threads that deflate the members of a .mod / .ref / .readset file (mg_pgzip.c) */
  long xferPieceKb;        /* MODGPU_XFER_PIECE_KB: dev, bytes per piece of the array transfers (default 4096) */
  long xferStreams;        /* MODGPU_XFER_STREAMS: dev, 0 = the array transfers' copies on the device's default stream instead of a stream per transfer thread */
  long xferThreads;        /* MODGPU_XFER_THREADS: host threads of the array transfers (mg_xfer.hip) */
  long seedTiming, uploadTiming, textTiming, parseTiming;   /* MODGPU_*_TIMING prints */
  long scanDebug, bucketDebug;                               /* only read by -DMG_ABLATE builds */
} MgKnobs;
const MgKnobs *mgKnobs (void);
void mgReloadKnobs (void);
/* the CPUs this process may really use: what is online, cut down by the affinity mask and by a cgroup CPU quota (cpu.max) -- the
   number the host-side thread teams size themselves by (worked out once) */
int mgCpuBudget (void);
#include <stddef.h>
void mgHugeHint (void *p, size_t n);      /* madvise (MADV_HUGEPAGE) on the whole pages of a large malloc ()ed block */
void *mgAllocBig (size_t n);              /* malloc () + that hint */
#ifdef __cplusplus
}
#endif
#endif
