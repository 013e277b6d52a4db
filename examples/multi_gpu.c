/* examples/multi_gpu.c — BASELINE config 4 in plain C against include/modgpu.h: a synthetic ONT-like read set in contiguous blocks, one
 * block per GPU; every GPU scans its block and builds its own modset (no collective on the data path); the global depth histogram is
 * one RCCL all-reduce of 65 536 x U64 (mgHistogramAllReduce); and, for the exact global set, the per-GPU modsets are folded into GPU
 * 0's in rank order (mgModsetMergeRankOrder: modsetMerge semantics, modset.c:106-128), which must reproduce -- bit for bit -- the
 * modset ONE stream over all the blocks builds (first-occurrence indices, saturated depths).  Third exchange of SURVEY 8(e): reads COUNTED
 * against one fixed set that every GPU holds (modasm's ingest, modasm.c:151-191: here the genome's own set, depth zeroed) -- every GPU counts
 * its block (mgReadsetRead), the counts are summed over the GPUs and clamped (mgDepthAllReduce), and must equal the counts of one stream
 * over all the blocks.  One process, one host thread per GPU.
 *
 *   gcc -O2 -pthread -I include examples/multi_gpu.c -o multi_gpu -L modimizer_amd -lmodgpu -lm \
 *       -Wl,-rpath,$PWD/modimizer_amd -Wl,-rpath,/opt/rocm/lib
 *   ./multi_gpu [nGpus (default: all)] [Mbp per GPU (default 200)] [check: 1 = also build the single-stream set and compare (default 1)]
 *
 * Prints one line per check and "MULTI_GPU_OK" when every one holds; exit code 0 then.
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "modgpu.h"

#define K 21
#define D 64
#define SEED 17
#define BITS 28

static U64 mix64 (U64 *s) { U64 z = (*s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static double unif (U64 *s) { return (double) (mix64 (s) >> 11) * (1.0 / 9007199254740992.0); }

/* a block's reads: log-normal lengths (N50 20 kb: sigma 0.6, mu = ln 20000 - sigma^2, clamped to [500, 200000]), uniform starts, random strands */
typedef struct { U32 n; U64 total; U64 *start, *off; U8 *strand; } Plan;
static Plan makePlan (U64 bases, U64 genome, U64 seed)
{
  Plan p; memset (&p, 0, sizeof (p));
  size_t cap = (size_t) (bases / 4000 + 1024);
  p.start = (U64 *) malloc (cap * 8); p.off = (U64 *) malloc ((cap + 1) * 8); p.strand = (U8 *) malloc (cap);
  const double sigma = 0.6, mu = log (20000.0) - sigma * sigma;
  U64 s = seed, tot = 0;
  while (tot < bases && p.n < cap)
    { const double u1 = unif (&s) + 1e-300, u2 = unif (&s);
      double len = exp (mu + sigma * sqrt (-2.0 * log (u1)) * cos (6.283185307179586 * u2));
      if (len < 500) len = 500;
      if (len > 200000) len = 200000;
      U64 L = (U64) len; if (L > genome) L = genome;
      if (tot + L > bases) L = bases - tot;
      if (L < 32) break;
      p.off[p.n] = tot; p.start[p.n] = (U64) (unif (&s) * (double) (genome - L)); p.strand[p.n] = (U8) (mix64 (&s) & 1);
      tot += L; ++p.n;
    }
  p.off[p.n] = tot; p.total = tot;
  return p;
}

typedef struct
{ int rank, nGpus; MgComm *comm; U64 blockBases, genomeBases; int check;
  Modset *ms; Seqhash *sh; U64 nHash; U64 hist[65536], local[65536]; double buildMs, reduceMs; int ok;
  void *dReads, *dOff; Plan plan;                 /* kept for the single-stream check on GPU 0 */
  Modset *fixed;                                  /* the genome's set with this GPU's hit counts, then everybody's */
  pthread_barrier_t *bar;
} Worker;

/* (a rank that has failed skips what follows but still meets the others at the barriers) */
#define CK(call) do { if (w->ok) { MgStatus s_ = (call); if (s_) { fprintf (stderr, "rank %d: %s failed: %s\n", w->rank, #call, mgLastError ()); w->ok = 0; } } } while (0)
static double nowMs (void) { struct timespec t; clock_gettime (CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

/* the block of `rank` on the calling thread's device: genome, reads */
static int makeBlock (Worker *w, int rank, void **dReads, void **dOff, Plan *plan)
{
  void *dGenome = 0, *dStart = 0, *dStrand = 0;
  *plan = makePlan (w->blockBases, w->genomeBases, 1000 + (U64) rank);
  if (mgDeviceAlloc (&dGenome, mgPackedWords (w->genomeBases) * 4) || mgSynthGenome ((U32 *) dGenome, w->genomeBases, 12345, 0)) return -1;
  if (mgDeviceAlloc (dReads, mgPackedWords (plan->total) * 4) || mgDeviceAlloc (dOff, ((size_t) plan->n + 1) * 8)
      || mgDeviceAlloc (&dStart, ((size_t) plan->n + 1) * 8) || mgDeviceAlloc (&dStrand, (size_t) plan->n + 16)) return -1;
  if (mgMemcpyH2D (*dOff, plan->off, ((size_t) plan->n + 1) * 8, 0) || mgMemcpyH2D (dStart, plan->start, (size_t) plan->n * 8, 0)
      || mgMemcpyH2D (dStrand, plan->strand, plan->n, 0)) return -1;
  if (mgSynthReads ((const U32 *) dGenome, w->genomeBases, (const U64 *) dStart, (const U64 *) *dOff, (const U8 *) dStrand, plan->n, plan->total,
                    0.05, 777 + (U64) rank, (U32 *) *dReads, 0) || mgStreamSynchronize (0)) return -1;
  mgDeviceFree (dGenome); mgDeviceFree (dStart); mgDeviceFree (dStrand);
  return 0;
}

/* the genome's own modset on the calling thread's device, depth zeroed as modasm does before it counts (modasm.c:158): the same on every GPU */
static Modset *fixedSet (Worker *w, Seqhash *sh)
{
  void *dGenome = 0, *dOff = 0; U64 off[2] = { 0, w->genomeBases }, nh = 0;
  Modset *ms = modsetCreate (sh, BITS, 0);
  if (mgDeviceAlloc (&dGenome, mgPackedWords (w->genomeBases) * 4) || mgSynthGenome ((U32 *) dGenome, w->genomeBases, 12345, 0) || mgDeviceAlloc (&dOff, 16)
      || mgMemcpyH2D (dOff, off, 16, 0) || mgAddReadsDevice (ms, (const U32 *) dGenome, w->genomeBases, (const U64 *) dOff, 1, &nh, 0) || modsetSyncToHost (ms, 0)) return 0;
  memset (ms->depth, 0, ((size_t) ms->max + 1) * sizeof (U16)); mgModsetHostChanged (ms);
  mgDeviceFree (dGenome); mgDeviceFree (dOff);
  return ms;
}
/* a block as the host bytes mgReadsetRead takes (one base a byte), appended at bases + at */
static int blockBytes (const void *dReads, U64 total, char *bases)
{
  void *dB = 0;
  if (mgDeviceAlloc (&dB, total + 16) || mgUnpackDevice ((const U32 *) dReads, total, (U8 *) dB, 0) || mgMemcpyD2H (bases, dB, total, 0)) return -1;
  mgDeviceFree (dB);
  return 0;
}

static void *work (void *v)
{
  Worker *w = (Worker *) v;
  w->ok = 1;
  CK (mgSetDevice (w->rank));
  if (makeBlock (w, w->rank, &w->dReads, &w->dOff, &w->plan)) { fprintf (stderr, "rank %d: %s\n", w->rank, mgLastError ()); w->ok = 0; }
  w->sh = seqhashCreate (K, D, SEED);
  w->ms = modsetCreate (w->sh, BITS, 0);
  pthread_barrier_wait (w->bar);
  /* the step: scan + build on this GPU's block, then the histogram all-reduce */
  double t0 = nowMs ();
  CK (mgAddReadsDevice (w->ms, (const U32 *) w->dReads, w->plan.total, (const U64 *) w->dOff, w->plan.n, &w->nHash, 0));
  CK (mgStreamSynchronize (0));
  double t1 = nowMs ();
  /* (every rank must be able to take part in an exchange: one that failed would leave the others waiting in it) */
  pthread_barrier_wait (w->bar);
  int all = 1; for (int r = 0 ; r < w->nGpus ; ++r) all = all && w[r - w->rank].ok;
  double t1b = nowMs ();
  if (all) CK (mgHistogramAllReduce (w->ms, w->hist, w->comm));
  double t2 = nowMs () - (t1b - t1);
  w->buildMs = t1 - t0; w->reduceMs = t2 - t1;
  /* this GPU's own histogram, for the check that the all-reduce summed what it should */
  if (w->ok)
  { void *dH = 0; CK (mgDeviceAlloc (&dH, 65536 * 8)); CK (mgMemsetD (dH, 0, 65536 * 8, 0)); CK (modsetDepthHistogramDevice (w->ms, (U64 *) dH, 0));
    CK (mgMemcpyD2H (w->local, dH, 65536 * 8, 0)); mgDeviceFree (dH);
  }
  pthread_barrier_wait (w->bar);
  all = 1; for (int r = 0 ; r < w->nGpus ; ++r) all = all && w[r - w->rank].ok;
  if (w->check && all) CK (mgModsetMergeRankOrder (w->ms, w->comm, 0));
  /* reads counted against one fixed set: this GPU's block, then the sum over the GPUs */
  if (w->check && w->ok)
    { char *bases = (char *) malloc (w->plan.total + 1); int64_t *off = (int64_t *) malloc (((size_t) w->plan.n + 1) * sizeof (int64_t));
      for (U32 i = 0 ; i <= w->plan.n ; ++i) off[i] = (int64_t) w->plan.off[i];
      w->fixed = fixedSet (w, w->sh);
      if (!w->fixed || !bases || !off || blockBytes (w->dReads, w->plan.total, bases)) { fprintf (stderr, "rank %d: %s\n", w->rank, mgLastError ()); w->ok = 0; }
      else
        { MgReadset *rs = mgReadsetCreate (w->fixed);
          if (mgReadsetRead (rs, bases, off, (int) w->plan.n)) w->ok = 0;
          mgReadsetDestroy (rs);
        }
      free (bases); free (off);
    }
  pthread_barrier_wait (w->bar);
  all = 1; for (int r = 0 ; r < w->nGpus ; ++r) all = all && w[r - w->rank].ok;
  if (w->check && all) CK (mgDepthAllReduce (w->fixed, w->comm));
  return 0;
}

int main (int argc, char **argv)
{
  int nGpus = argc > 1 ? atoi (argv[1]) : mgDeviceCount ();
  const double mbp = argc > 2 ? atof (argv[2]) : 200;
  const int check = argc > 3 ? atoi (argv[3]) : 1;
  if (nGpus < 1 || nGpus > mgDeviceCount ()) { fprintf (stderr, "%d GPUs asked for, %d present (libmodgpu has no CPU fallback)\n", nGpus, mgDeviceCount ()); return 2; }
  MgComm **comms = (MgComm **) calloc ((size_t) nGpus, sizeof (MgComm *));
  if (mgCommInitAll (comms, nGpus, 0)) { fprintf (stderr, "mgCommInitAll: %s\n", mgLastError ()); return 1; }
  Worker *w = (Worker *) calloc ((size_t) nGpus, sizeof (Worker));
  pthread_t *th = (pthread_t *) calloc ((size_t) nGpus, sizeof (pthread_t));
  pthread_barrier_t bar; pthread_barrier_init (&bar, 0, (unsigned) nGpus);
  const U64 blockBases = (U64) (mbp * 1e6), genomeBases = blockBases * 8 / 30 > 1000000 ? blockBases * 8 / 30 : 1000000;   /* config 4: 8 blocks, 30x of the genome in all */
  for (int r = 0 ; r < nGpus ; ++r)
    { w[r].rank = r; w[r].nGpus = nGpus; w[r].comm = comms[r]; w[r].blockBases = blockBases; w[r].genomeBases = genomeBases; w[r].check = check; w[r].bar = &bar;
      pthread_create (&th[r], 0, work, &w[r]);
    }
  for (int r = 0 ; r < nGpus ; ++r) pthread_join (th[r], 0);
  int ok = 1;
  for (int r = 0 ; r < nGpus ; ++r) ok = ok && w[r].ok;
  if (!ok) { fprintf (stderr, "a rank failed\n"); return 1; }

  /* 1. the all-reduced histogram is the sum of the ranks' own, on every rank, and counts every rank's entries */
  U64 bases = 0, hashes = 0; double slowest = 0;
  for (int r = 0 ; r < nGpus ; ++r) { bases += w[r].plan.total; hashes += w[r].nHash; if (w[r].buildMs + w[r].reduceMs > slowest) slowest = w[r].buildMs + w[r].reduceMs; }
  int histOk = 1; U64 histEntries = 0;
  for (int b = 0 ; b < 65536 ; ++b)
    { U64 sum = 0; for (int r = 0 ; r < nGpus ; ++r) sum += w[r].local[b];
      for (int r = 0 ; r < nGpus ; ++r) if (w[r].hist[b] != sum) histOk = 0;
      histEntries += sum;
    }
  /* (entries per rank BEFORE the merge: the histogram's total) */
  printf ("%d GPU(s), %.3f Gbp in all, %llu modimizers: scan + build + all-reduce %.2f ms on the slowest rank = %.1f Gbp/s (all-reduce %.3f ms on rank 0)\n",
          nGpus, bases / 1e9, (unsigned long long) hashes, slowest, bases / slowest / 1e6, w[0].reduceMs);
  printf ("histogram: all-reduced == sum of the ranks' own on every rank: %s (%llu entries over all ranks)\n", histOk ? "yes" : "NO", (unsigned long long) histEntries);
  ok = ok && histOk && histEntries > 0;

  /* 2. the merge in rank order == the set one stream over all the blocks builds (GPU 0 builds that one now) */
  if (check)
    { mgSetDevice (0);
      Seqhash *sh = seqhashCreate (K, D, SEED); Modset *one = modsetCreate (sh, BITS, 0);
      for (int r = 0 ; r < nGpus && ok ; ++r)
        { void *dR = w[r].dReads, *dO = w[r].dOff; Plan pl = w[r].plan;
          if (r) { if (makeBlock (&w[0], r, &dR, &dO, &pl)) { fprintf (stderr, "%s\n", mgLastError ()); ok = 0; break; } }      /* block r again, on GPU 0 */
          U64 nh = 0;
          if (mgAddReadsDevice (one, (const U32 *) dR, pl.total, (const U64 *) dO, pl.n, &nh, 0)) { fprintf (stderr, "%s\n", mgLastError ()); ok = 0; }
          if (r) { mgDeviceFree (dR); mgDeviceFree (dO); free (pl.start); free (pl.off); free (pl.strand); }
        }
      if (ok && (modsetSyncToHost (one, 0) || modsetSyncToHost (w[0].ms, 0))) { fprintf (stderr, "%s\n", mgLastError ()); ok = 0; }
      if (ok)
        { Modset *m = w[0].ms;
          int same = m->max == one->max && !memcmp (m->value + 1, one->value + 1, (size_t) one->max * 8) && !memcmp (m->depth + 1, one->depth + 1, (size_t) one->max * 2);
          printf ("merge in rank order: %u entries; identical to the single-stream build over all blocks (value[], depth[]): %s\n", m->max, same ? "yes" : "NO");
          ok = ok && same;
        }
      modsetDestroy (one);
      /* 3. counts against the fixed set, summed over the GPUs == one stream over all the blocks counting into the same set */
      if (ok)
        { U64 tot = 0, nr = 0; for (int r = 0 ; r < nGpus ; ++r) { tot += w[r].plan.total; nr += w[r].plan.n; }
          char *bases = (char *) malloc (tot + 1); int64_t *off = (int64_t *) malloc ((nr + 1) * sizeof (int64_t));
          U64 at = 0, ri = 0;
          for (int r = 0 ; r < nGpus && ok ; ++r)
            { void *dR = w[r].dReads, *dO = w[r].dOff; Plan pl = w[r].plan;
              if (r) { if (makeBlock (&w[0], r, &dR, &dO, &pl)) { fprintf (stderr, "%s\n", mgLastError ()); ok = 0; break; } }
              if (blockBytes (dR, pl.total, bases + at)) { fprintf (stderr, "%s\n", mgLastError ()); ok = 0; }
              for (U32 i = 0 ; i < pl.n ; ++i) off[ri++] = (int64_t) (at + pl.off[i]);
              at += pl.total;
              if (r) { mgDeviceFree (dR); mgDeviceFree (dO); free (pl.start); free (pl.off); free (pl.strand); }
            }
          off[ri] = (int64_t) at;
          Modset *whole = ok ? fixedSet (&w[0], sh) : 0;
          if (ok && whole)
            { MgReadset *rs = mgReadsetCreate (whole);
              ok = ok && mgReadsetRead (rs, bases, off, (int) nr) == 0;
              mgReadsetDestroy (rs);
              int same = 1; U64 hits = 0;
              for (int r = 0 ; r < nGpus ; ++r)
                same = same && w[r].fixed->max == whole->max && !memcmp (w[r].fixed->depth + 1, whole->depth + 1, (size_t) whole->max * 2);
              for (U32 i = 1 ; i <= whole->max ; ++i) hits += whole->depth[i];
              printf ("reads counted against the genome's set (%u entries, %llu hits): sum over the GPUs == one stream, on every GPU: %s\n", whole->max, (unsigned long long) hits, same ? "yes" : "NO");
              ok = ok && same && hits > 0;
              modsetDestroy (whole);
            }
          else ok = 0;
          free (bases); free (off);
        }
      mgSeqhashDestroy (sh);
    }
  for (int r = 0 ; r < nGpus ; ++r)
    { mgSetDevice (r); modsetDestroy (w[r].ms); if (w[r].fixed) modsetDestroy (w[r].fixed); mgSeqhashDestroy (w[r].sh); mgDeviceFree (w[r].dReads); mgDeviceFree (w[r].dOff); mgCommDestroy (comms[r]); }
  if (ok) printf ("MULTI_GPU_OK\n");
  return ok ? 0 : 1;
}
