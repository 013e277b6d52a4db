/* oracle/cpu_bench.c — TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg; never linked into the product).
 *
 * The CPU path of SURVEY §8(d)(2): the sample's reads sharded in contiguous blocks over every host core
 * this process may use, each thread building a PRIVATE modset with the oracle's restatement of
 * modRCiterator/modRCnext + modsetIndexFind + the saturating depth bump (seqhash.c:154-196, modset.c:45-62,
 * modutils.c:19-31), and the private sets then merged IN BLOCK ORDER with modsetMerge semantics
 * (modset.c:106-128: second set's values inserted in its index order, depth = min (65535, a+b)).  The merge is a
 * binary tree over adjacent blocks, its levels run in parallel, so the block order — and with it the
 * single-stream first-occurrence index order — is kept.  The merged set is then compared, entry for entry,
 * with a single-thread build of the same sample (untimed), so the figure is for a correct result.
 *
 * Beside it, what the host can do at best with the same algorithm ("scan_then_insert"): the scan (seqhash.c:154-196) is
 * the part that parallelises freely, so every thread scans its block into a private k-mer list, and ONE thread then makes
 * the single ordered insert stream of modset.c:45-62 + modutils.c:26 over the lists in block order, prefetching the
 * index[] slot of the k-mer sixteen ahead (the k-mers are all known).  No merge; the same result by construction, and
 * checked.  The faster of the two is reported as "best".
 *
 * usage: cpu_bench <sample.bin> k d seed bits     (sample.bin: u64 nReads, u64 nBases, i64 off[nReads+1], bases)
 * prints one JSON line.
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <sched.h>
#include <time.h>
#include <unistd.h>
#include "oracle.h"

static double now (void) { struct timespec t; clock_gettime (CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static int bitsFor (uint64_t entries)
{ int b = 20; while ((((uint64_t) 1 << b) >> 2) - 1 <= entries + 16 && b < 34) ++b; return b; }

typedef struct {
  const OrcHasher *h; const uint8_t *bases; const int64_t *off; int64_t r0, r1; int d;
  OrcModset *ms; int64_t hashes;
} Shard;

static void *buildShard (void *arg)
{
  Shard *s = (Shard *) arg;
  uint64_t nb = (uint64_t) (s->off[s->r1] - s->off[s->r0]);
  s->ms = orcModsetCreate (s->h, bitsFor (nb / (uint64_t) s->d + nb / (uint64_t) (4 * s->d) + 1024), 0);
  s->hashes = orcScanMany (s->h, s->bases, s->off + s->r0, s->r1 - s->r0, s->ms);
  return 0;
}

/* scan only: the block's modimizers, in order, into a private list */
typedef struct { const OrcHasher *h; const uint8_t *bases; const int64_t *off; int64_t r0, r1; int d; uint64_t *km; int64_t n, cap; } ScanJob;
static void *scanShard (void *arg)
{
  ScanJob *s = (ScanJob *) arg;
  uint64_t nb = (uint64_t) (s->off[s->r1] - s->off[s->r0]);
  s->cap = (int64_t) (nb / (uint64_t) s->d + nb / (uint64_t) (2 * s->d) + 4096);
  s->km = (uint64_t *) malloc ((size_t) s->cap * 8);
  s->n = 0;
  for (int64_t r = s->r0 ; r < s->r1 ; ++r)
    { const int64_t len = s->off[r + 1] - s->off[r];
      int64_t got = orcScanRead (s->h, s->bases + s->off[r], len, s->km + s->n, 0, 0, s->cap - s->n);
      if (got > s->cap - s->n)                          /* a denser block than expected: grow and redo this read */
        { s->cap = 2 * (s->n + got) + 4096; s->km = (uint64_t *) realloc (s->km, (size_t) s->cap * 8);
          got = orcScanRead (s->h, s->bases + s->off[r], len, s->km + s->n, 0, 0, s->cap - s->n);
        }
      s->n += got;
    }
  return 0;
}

typedef struct { const OrcHasher *h; OrcModset **left, *right; int ok; } Pair;
static void *mergePair (void *arg)
{
  Pair *p = (Pair *) arg;
  OrcModset *a = *p->left, *b = p->right;
  p->ok = 1;
  if ((uint64_t) a->max + b->max + 1 >= (a->tableSize >> 2))          /* a's table too small for both: regrow by re-insertion */
    { OrcModset *big = orcModsetCreate (p->h, bitsFor ((uint64_t) a->max + b->max), 0);
      p->ok = orcModsetMerge (big, a);
      orcModsetDestroy (a);
      *p->left = a = big;
    }
  if (p->ok) p->ok = orcModsetMerge (a, b);
  orcModsetDestroy (b);
  return 0;
}

int main (int argc, char **argv)
{
  if (argc < 6) { fprintf (stderr, "usage: %s sample.bin k d seed bits\n", argv[0]); return 2; }
  FILE *f = fopen (argv[1], "rb");
  if (!f) { fprintf (stderr, "cannot open %s\n", argv[1]); return 2; }
  uint64_t nReads, total;
  if (fread (&nReads, 8, 1, f) != 1 || fread (&total, 8, 1, f) != 1) return 2;
  int64_t *off = (int64_t *) malloc ((nReads + 1) * sizeof (int64_t));
  uint8_t *bases = (uint8_t *) malloc (total + 1);
  if (fread (off, 8, nReads + 1, f) != nReads + 1 || fread (bases, 1, total, f) != total) return 2;
  fclose (f);
  const int k = atoi (argv[2]), d = atoi (argv[3]), seed = atoi (argv[4]), bits = atoi (argv[5]);
  OrcHasher h;
  if (orcHasherInit (&h, k, d, seed)) return 2;

  /* how many threads: the cores this process may run on, capped by a cgroup CPU quota if there is one */
  const long online = sysconf (_SC_NPROCESSORS_ONLN);
  cpu_set_t set; int affinity = (int) online;
  if (sched_getaffinity (0, sizeof (set), &set) == 0) affinity = CPU_COUNT (&set);
  long quota = 0;
  { FILE *q = fopen ("/sys/fs/cgroup/cpu.max", "r");
    if (q) { char a[64]; long per = 0; if (fscanf (q, "%63s %ld", a, &per) == 2 && strcmp (a, "max") && per > 0) quota = (atol (a) + per - 1) / per; fclose (q); }
  }
  int T = affinity;
  if (quota > 0 && quota < T) T = (int) quota;
  if (getenv ("MODGPU_CPU_THREADS")) T = atoi (getenv ("MODGPU_CPU_THREADS"));
  if (T < 1) T = 1;
  if ((uint64_t) T > nReads) T = (int) nReads;

  /* contiguous blocks of reads, balanced by bases */
  Shard *sh = (Shard *) calloc ((size_t) T, sizeof (Shard));
  { int64_t r = 0;
    for (int t = 0 ; t < T ; ++t)
      { int64_t goal = (int64_t) ((__int128) total * (t + 1) / T);
        sh[t].h = &h; sh[t].bases = bases; sh[t].off = off; sh[t].d = d; sh[t].r0 = r;
        while (r < (int64_t) nReads && (off[r + 1] <= goal || t == T - 1)) ++r;
        if (t == T - 1) r = (int64_t) nReads;
        sh[t].r1 = r;
      }
  }
  pthread_t *th = (pthread_t *) malloc ((size_t) T * sizeof (pthread_t));
  double t0 = now ();
  for (int t = 0 ; t < T ; ++t) pthread_create (&th[t], 0, buildShard, &sh[t]);
  for (int t = 0 ; t < T ; ++t) pthread_join (th[t], 0);
  double tBuild = now () - t0;
  int64_t hashes = 0;
  for (int t = 0 ; t < T ; ++t) hashes += sh[t].hashes;

  /* tree merge of adjacent blocks: level by level, the pairs of a level in parallel */
  OrcModset **set_ = (OrcModset **) malloc ((size_t) T * sizeof (OrcModset *));
  for (int t = 0 ; t < T ; ++t) set_[t] = sh[t].ms;
  Pair *pr = (Pair *) calloc ((size_t) T, sizeof (Pair));
  int ok = 1;
  t0 = now ();
  for (int step = 1 ; step < T ; step *= 2)
    { int np = 0;
      for (int t = 0 ; t + step < T ; t += 2 * step)
        { pr[np].h = &h; pr[np].left = &set_[t]; pr[np].right = set_[t + step]; pthread_create (&th[np], 0, mergePair, &pr[np]); ++np; }
      for (int i = 0 ; i < np ; ++i) { pthread_join (th[i], 0); ok &= pr[i].ok; }
    }
  double tMerge = now () - t0;
  OrcModset *merged = set_[0];

  /* the check (untimed): one thread, one stream */
  OrcModset *one = orcModsetCreate (&h, bits, 0);
  t0 = now ();
  int64_t hashes1 = orcScanMany (&h, bases, off, (int64_t) nReads, one);
  double tOne = now () - t0;
  int same = ok && hashes1 == hashes && one->max == merged->max
             && !memcmp (one->value + 1, merged->value + 1, (size_t) one->max * 8)
             && !memcmp (one->depth + 1, merged->depth + 1, (size_t) one->max * 2);
  /* ---- scan in parallel, then one ordered insert stream ---- */
  ScanJob *sj = (ScanJob *) calloc ((size_t) T, sizeof (ScanJob));
  for (int t = 0 ; t < T ; ++t) { sj[t].h = &h; sj[t].bases = bases; sj[t].off = off; sj[t].d = d; sj[t].r0 = sh[t].r0; sj[t].r1 = sh[t].r1; }
  OrcModset *two = orcModsetCreate (&h, bits, 0);
  t0 = now ();
  for (int t = 0 ; t < T ; ++t) pthread_create (&th[t], 0, scanShard, &sj[t]);
  for (int t = 0 ; t < T ; ++t) pthread_join (th[t], 0);
  double tScan = now () - t0;
  t0 = now ();
  int64_t hashes2 = 0;
  for (int t = 0 ; t < T ; ++t)
    { const uint64_t *km = sj[t].km; const int64_t n = sj[t].n;
      for (int64_t i = 0 ; i < n ; ++i)
        { if (i + 16 < n) __builtin_prefetch (&two->index[orcHash (&h, km[i + 16]) & two->tableMask]);
          uint32_t ix = orcModsetFind (two, km[i], 1);                       /* modset.c:45-62 */
          uint16_t *di = &two->depth[ix]; ++*di; if (!*di) *di = 0xffff;     /* modutils.c:26 */
        }
      hashes2 += n;
    }
  double tInsert = now () - t0;
  int same2 = !two->overflow && hashes2 == hashes1 && one->max == two->max
              && !memcmp (one->value + 1, two->value + 1, (size_t) one->max * 8)
              && !memcmp (one->depth + 1, two->depth + 1, (size_t) one->max * 2);
  const double gMerge = total / (tBuild + tMerge) / 1e9, gInsert = total / (tScan + tInsert) / 1e9;
  char qs[32]; if (quota) snprintf (qs, sizeof (qs), "%ld", quota); else strcpy (qs, "null");
  printf ("{\"value\": %.4f, \"unit\": \"Gbp/s\", \"threads\": %d, \"kind\": \"port\", \"cores_online\": %ld, \"affinity_cores\": %d, "
          "\"cgroup_cpu_quota\": %s, \"build_s\": %.3f, \"merge_s\": %.3f, \"merge\": \"private per-thread modsets merged in block order "
          "(modsetMerge semantics, binary tree over adjacent blocks)\", \"entries\": %u, \"hashes\": %lld, "
          "\"equals_single_thread_build\": %s, \"single_thread_port_gbps\": %.4f, "
          "\"scan_then_insert\": {\"value\": %.4f, \"scan_s\": %.3f, \"insert_s\": %.3f, \"what\": \"%d threads scan their blocks into k-mer lists, one thread "
          "makes the single ordered insert stream over them (index slot prefetched 16 ahead); no merge\", \"equals_single_thread_build\": %s}, "
          "\"best\": {\"value\": %.4f, \"which\": \"%s\"}}\n",
          gMerge, T, online, affinity, qs,
          tBuild, tMerge, merged->max, (long long) hashes, same ? "true" : "false", total / tOne / 1e9,
          gInsert, tScan, tInsert, T, same2 ? "true" : "false",
          gInsert > gMerge ? gInsert : gMerge, gInsert > gMerge ? "scan_then_insert" : "merge_tree");
  same = same && same2;
  return same ? 0 : 1;
}
