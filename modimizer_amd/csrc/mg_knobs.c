/* mg_knobs.c — the library's environment knobs, read once (see mg_knobs.h) */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/mman.h>
#include "mg_knobs.h"

static MgKnobs gKnobs;
static pthread_once_t gOnce = PTHREAD_ONCE_INIT;

static long num (const char *name)
{ const char *e = getenv (name); return e && *e ? atol (e) : MG_KNOB_UNSET; }

static void readAll (void)
{
  MgKnobs k;
  const char *e = getenv ("MODGPU_TABLE_PATH");
  k.tablePath = e && *e ? (long) e[0] : MG_KNOB_UNSET;
  e = getenv ("MODGPU_FIND_PATH");
  k.findPath = e && *e ? (long) e[0] : MG_KNOB_UNSET;
  k.partPacked = num ("MODGPU_PART_PACKED");       k.partBig = num ("MODGPU_PART_BIG");
  k.addChunk = num ("MODGPU_ADD_CHUNK");           k.scanGrid = num ("MODGPU_SCAN_GRID");
  k.scanGeneric = num ("MODGPU_SCAN_GENERIC");     k.scanHist = num ("MODGPU_SCAN_HIST");
  k.scanDiv64 = num ("MODGPU_SCAN_DIV64");        k.minTiled = num ("MODGPU_MIN_TILED");
  k.minTile = num ("MODGPU_MIN_TILE");
  k.noSegmentInput = num ("MODGPU_NO_SEGMENT_INPUT");
  k.rankSliceShift = num ("MODGPU_RANK_SLICE_SHIFT");
  k.flagPolarity = num ("MODGPU_FLAG_POLARITY");   k.mergeSlots = num ("MODGPU_MERGE_SLOTS");
  k.bucketR = num ("MODGPU_BUCKET_R");             k.bucketT = num ("MODGPU_BUCKET_T");
  e = getenv ("MODGPU_HOT_SPLIT");
  k.hotSplit = e && *e ? atol (e) : MG_KNOB_UNSET;
  { const char *c = e ? strchr (e, ',') : 0; k.hotChunk = c ? atol (c + 1) : MG_KNOB_UNSET; }
  k.noAvx2 = num ("MODGPU_NO_AVX2");               k.textHost = num ("MODGPU_TEXT_HOST");
  k.textWindowKb = num ("MODGPU_TEXT_WINDOW_KB");
  k.fileBatchMbp = num ("MODGPU_FILE_BATCH_MBP");  k.fileBatchBases = num ("MODGPU_FILE_BATCH_BASES");
  k.queryHostChain = num ("MODGPU_QUERY_HOST_CHAIN");
  k.iterHostBelow = num ("MODGPU_ITER_HOST_BELOW");
  k.segSlack = num ("MODGPU_SEG_SLACK");
  k.partDigits = num ("MODGPU_PART_DIGITS");
  k.findBits = num ("MODGPU_FIND_BITS");
  k.scatterGrid = num ("MODGPU_SCATTER_GRID");     k.tableLoad = num ("MODGPU_TABLE_LOAD");
  k.tightLoad = num ("MODGPU_TIGHT_LOAD");         k.find8 = num ("MODGPU_FIND8");
  k.packThreads = num ("MODGPU_PACK_THREADS");     k.parseThreads = num ("MODGPU_PARSE_THREADS");
  k.xferThreads = num ("MODGPU_XFER_THREADS");       k.gzipThreads = num ("MODGPU_GZIP_THREADS");
  k.xferPieceKb = num ("MODGPU_XFER_PIECE_KB");      k.xferStreams = num ("MODGPU_XFER_STREAMS");
  k.seedTiming = num ("MODGPU_SEED_TIMING");       k.uploadTiming = num ("MODGPU_UPLOAD_TIMING");
  k.textTiming = num ("MODGPU_TEXT_TIMING");       k.parseTiming = getenv ("MODGPU_PARSE_TIMING") ? 1 : MG_KNOB_UNSET;
  k.scanDebug = num ("MODGPU_SCAN_DEBUG");         k.bucketDebug = num ("MODGPU_BUCKET_DEBUG");
  gKnobs = k;
}

const MgKnobs *mgKnobs (void) { pthread_once (&gOnce, readAll); return &gKnobs; }
void mgIterKnobsReloaded (void) __attribute__ ((weak));      /* mg_host.c: a value cached from a knob */
void mgReloadKnobs (void) { pthread_once (&gOnce, readAll); readAll (); if (mgIterKnobsReloaded) mgIterKnobsReloaded (); }

static int gCpuBudget;
static pthread_once_t gCpuOnce = PTHREAD_ONCE_INIT;
static void cpuBudgetOnce (void)
{
  long v = sysconf (_SC_NPROCESSORS_ONLN);
  cpu_set_t set;
  if (sched_getaffinity (0, sizeof (set), &set) == 0 && CPU_COUNT (&set) < v) v = CPU_COUNT (&set);
  FILE *q = fopen ("/sys/fs/cgroup/cpu.max", "r");           /* "max 100000" or "<quota> <period>" */
  if (q)
    { char a[64]; long per = 0;
      if (fscanf (q, "%63s %ld", a, &per) == 2 && strcmp (a, "max") && per > 0) { const long c = (atol (a) + per - 1) / per; if (c < v) v = c; }
      fclose (q);
    }
  gCpuBudget = v < 1 ? 1 : (int) v;
}
int mgCpuBudget (void) { pthread_once (&gCpuOnce, cpuBudgetOnce); return gCpuBudget; }

/* arrays of hundreds of megabytes that a transfer from the device writes end to end: with 4 KiB pages that is a page fault per
   4 KiB (a million for a 4 GiB index[]), with transparent huge pages one per 2 MiB.  A hint, not a requirement (the kernel's THP
   mode decides); the memory stays what malloc () returned, free ()-able as the reference's callers free it. */
void mgHugeHint (void *p, size_t n)
{
#ifdef MADV_HUGEPAGE
  if (p && n >= ((size_t) 8 << 20))
    { const size_t pg = 4096;
      char *a = (char *) (((size_t) p + pg - 1) & ~(pg - 1)), *e = (char *) (((size_t) p + n) & ~(pg - 1));
      if (e > a) (void) madvise (a, (size_t) (e - a), MADV_HUGEPAGE);
    }
#else
  (void) p; (void) n;
#endif
}
void *mgAllocBig (size_t n) { void *p = malloc (n ? n : 1); mgHugeHint (p, n); return p; }
