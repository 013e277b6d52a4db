/* oracle/orc_readset.c — TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * CPU restatement of modasm's read ingest: readsetFileRead (modasm.c:151-191), invBuild
 * (modasm.c:258-287), readsetStats (modasm.c:193-253) and the .readset stream of readsetWrite
 * (modasm.c:108-126), one read at a time as the reference does it.  Pinned against files and output
 * of the reference program itself (tests/golden/asm_*).
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define TOPBIT  0x80000000u        /* modasm.c:22 */
#define TOPMASK 0x7fffffffu

OrcReadset *orcReadsetCreate (OrcModset *ms)                     /* modasm.c:90-98; entry 0 of reads is burnt */
{
  OrcReadset *rs = (OrcReadset *) calloc (1, sizeof (OrcReadset));
  rs->ms = ms;
  rs->capReads = 1024;
  rs->reads = (OrcRead *) calloc ((size_t) rs->capReads, sizeof (OrcRead));
  rs->hitStart = (uint64_t *) calloc ((size_t) rs->capReads + 1, sizeof (uint64_t));
  return rs;
}

void orcReadsetDestroy (OrcReadset *rs)
{
  if (!rs) return;
  free (rs->reads); free (rs->hitStart); free (rs->hit); free (rs->dx); free (rs->invStart); free (rs->invSpace); free (rs);
}

void orcReadsetBegin (OrcReadset *rs)                             /* modasm.c:158: depth is rebuilt from this file */
{ memset (rs->ms->depth, 0, ((size_t) rs->ms->max + 1) * sizeof (uint16_t)); }

void orcReadsetAddRead (OrcReadset *rs, const uint8_t *s, int64_t len)          /* modasm.c:161-188 */
{
  OrcModset *ms = rs->ms;
  if (rs->nReads + 2 > rs->capReads)
    { rs->capReads *= 2;
      rs->reads = (OrcRead *) realloc (rs->reads, (size_t) rs->capReads * sizeof (OrcRead));
      memset (rs->reads + rs->capReads / 2, 0, (size_t) (rs->capReads / 2) * sizeof (OrcRead));
      rs->hitStart = (uint64_t *) realloc (rs->hitStart, ((size_t) rs->capReads + 1) * sizeof (uint64_t));
    }
  OrcRead *rd = &rs->reads[++rs->nReads];
  memset (rd, 0, sizeof (*rd));
  rd->len = (int32_t) len;
  int64_t cap = len > 0 ? len : 1;
  uint64_t *kmer = (uint64_t *) malloc ((size_t) cap * sizeof (uint64_t));
  int32_t *pos = (int32_t *) malloc ((size_t) cap * sizeof (int32_t));
  uint8_t *isF = (uint8_t *) malloc ((size_t) cap);
  int64_t n = orcScanRead (&ms->hasher, s, len, kmer, pos, isF, cap);
  if (rs->totHit + (uint64_t) n + 1 > rs->capHit)
    { rs->capHit = (rs->totHit + (uint64_t) n + 1) * 2;
      rs->hit = (uint32_t *) realloc (rs->hit, rs->capHit * sizeof (uint32_t));
      rs->dx = (uint16_t *) realloc (rs->dx, rs->capHit * sizeof (uint16_t));
    }
  rs->hitStart[rs->nReads] = rs->totHit;
  int lastPos = 0;
  for (int64_t i = 0 ; i < n ; ++i)
    { uint32_t index = orcModsetFind (ms, kmer[i], 0);
      if (index)
        { rs->hit[rs->totHit] = isF[i] ? (index | TOPBIT) : index;
          rs->dx[rs->totHit] = (uint16_t) (pos[i] - lastPos); lastPos = pos[i];
          ++rs->totHit; ++rd->nHit;
          uint16_t *di = &ms->depth[index]; ++*di; if (!*di) *di = 0xffff;     /* modasm.c:174 */
        }
      else ++rd->nMiss;
    }
  rs->hitStart[rs->nReads + 1] = rs->totHit;
  free (kmer); free (pos); free (isF);
}

void orcReadsetFinish (OrcReadset *rs)                            /* invBuild, modasm.c:258-287 */
{
  OrcModset *ms = rs->ms;
  free (rs->invStart); free (rs->invSpace);
  rs->invStart = (uint64_t *) calloc ((size_t) ms->max + 2, sizeof (uint64_t));
  rs->invSpace = (uint32_t *) malloc ((rs->totHit ? rs->totHit : 1) * sizeof (uint32_t));
  uint64_t off = 0;
  for (uint32_t i = 1 ; i <= ms->max ; ++i)
    { rs->invStart[i] = off;
      if (ms->depth[i] && ms->depth[i] < 0xffff) off += ms->depth[i];
    }
  rs->invStart[ms->max + 1] = off;
  uint64_t *fill = (uint64_t *) malloc (((size_t) ms->max + 2) * sizeof (uint64_t));
  memcpy (fill, rs->invStart, ((size_t) ms->max + 2) * sizeof (uint64_t));
  for (int i = 1 ; i <= rs->nReads ; ++i)
    { OrcRead *rd = &rs->reads[i];
      for (int j = 0 ; j < 4 ; ++j) rd->nCopy[j] = 0;
      for (uint64_t h = rs->hitStart[i] ; h < rs->hitStart[i + 1] ; ++h)
        { uint32_t y = rs->hit[h] & TOPMASK;
          ++rd->nCopy[ms->info[y] & 3];
          if (ms->depth[y] < 0xffff) rs->invSpace[fill[y]++] = (uint32_t) i;
        }
    }
  free (fill);
}

void orcReadsetStats (const OrcReadset *rs, FILE *f)              /* modasm.c:193-253, after its modsetSummary call */
{
  const OrcModset *ms = rs->ms;
  uint32_t n = (uint32_t) rs->nReads;
  int nUnique0 = 0, nUnique1 = 0;
  uint64_t totLen = 0, totMiss = 0, lenUnique0 = 0, lenUnique1 = 0, totCopy[4] = { 0, 0, 0, 0 };
  for (uint32_t i = 1 ; i <= n ; ++i)
    { const OrcRead *rd = &rs->reads[i];
      totLen += (uint64_t) rd->len; totMiss += (uint64_t) rd->nMiss;
      for (int j = 0 ; j < 4 ; ++j) totCopy[j] += (uint64_t) rd->nCopy[j];
      if (rd->nCopy[1] == 0) { ++nUnique0; lenUnique0 += (uint64_t) rd->len; }
      else if (rd->nCopy[1] == 1) { ++nUnique1; lenUnique1 += (uint64_t) rd->len; }
    }
  fprintf (f, "RS %d sequences, total length %llu (av %.1f)\n", n, (unsigned long long) totLen, totLen / (double) n);
  fprintf (f, "RS %llu mod hits, %.1f bp/hit, frac hit %.2f, av hits/read %.1f\n",
           (unsigned long long) rs->totHit, totLen / (double) rs->totHit, rs->totHit / (double) (totMiss + rs->totHit),
           rs->totHit / (double) n);
  fprintf (f, "RS hit distribution %.2f copy0, %.2f copy1, %.2f copy2, %.2f copyM\n",
           totCopy[0] / (double) rs->totHit, totCopy[1] / (double) rs->totHit,
           totCopy[2] / (double) rs->totHit, totCopy[3] / (double) rs->totHit);
  uint32_t nUniqueMulti = n - (uint32_t) nUnique0 - (uint32_t) nUnique1;
  fprintf (f, "RS num reads and av_len with 0 copy1 hits %d %.1f with 1 copy1 hits %d %.1f"
           " >1 copy1 hits %d %.1f av copy1 hits %.1f\n",
           nUnique0, lenUnique0 / (double) nUnique0, nUnique1, lenUnique1 / (double) nUnique1,
           nUniqueMulti, (totLen - lenUnique0 - lenUnique1) / (double) nUniqueMulti,
           (totCopy[1] - (uint64_t) nUnique1) / (double) nUniqueMulti);
  fprintf (f, "RS bad %u : %u repeat, %u order10, %u order1, ", 0u, 0u, 0u, 0u);       /* no read is marked before -b */
  fprintf (f, "%u no_match, %u low_hit, %u low_copy1\n", 0u, 0u, 0u);
  uint32_t nCopy[4] = { 0, 0, 0, 0 }, hitCopy[4] = { 0, 0, 0, 0 }, hit2Copy[4] = { 0, 0, 0, 0 };
  uint64_t depthCopy[4] = { 0, 0, 0, 0 };
  for (uint32_t i = 1 ; i <= ms->max ; ++i)
    { int j = ms->info[i] & 3;
      ++nCopy[j];
      if (ms->depth[i] > 0) ++hitCopy[j];
      if (ms->depth[i] > 1) { ++hit2Copy[j]; depthCopy[j] += ms->depth[i]; }
    }
  fprintf (f, "RS mod frac hit hit>1 av:");
  static const char *nm[4] = { "copy0", "copy1", "copy2", "copyM" };
  for (int j = 0 ; j < 4 ; ++j)
    fprintf (f, " %s %.3f %.3f %.1f", nm[j], hitCopy[j] / (double) nCopy[j], hit2Copy[j] / (double) nCopy[j],
             depthCopy[j] / (double) hit2Copy[j]);
  fprintf (f, "\n");
}

/* allocated elements of the reference's Array after appending elements 0..n-1 to
 * arrayCreate (first, size): array.c:144-170,180-183 */
static int arrayDimAfter (int first, int size, int n)
{
  int dim = first < 1 ? 1 : first;
  for (int i = 0 ; i < n ; ++i)
    if (i >= dim)
      { if ((long) dim * size < (1 << 23)) dim *= 2; else dim += 1024 + ((1 << 23) / size);
        if (i >= dim) dim = i + 1;
      }
  return dim;
}

/* the .readset stream (modasm.c:113-125), uncompressed; the two pointers inside every Read and the
 * one inside the Array header, which the reference dumps as they are in memory, are written as 0 */
int orcReadsetWrite (const OrcReadset *rs, FILE *f)
{
  struct { int32_t magic; int32_t pad0; uint64_t base; int32_t dim, size, max; int32_t pad1; } ah;      /* array.h:41-50 */
  struct { int32_t len, nHit; uint64_t hit, dx; uint8_t bad, other; uint16_t pad1; int32_t nMiss, contained, nCopy[4];
           uint32_t pad2[4]; uint32_t tail; } rec;                                                       /* modasm.c:30-57: 72 bytes */
  if (sizeof (ah) != 32 || sizeof (rec) != 72) return 0;
  if (fwrite ("RSMSHv2", 8, 1, f) != 1) return 0;
  if (fwrite (&rs->totHit, sizeof (uint64_t), 1, f) != 1) return 0;
  memset (&ah, 0, sizeof (ah));
  ah.magic = 8918274; ah.size = 72; ah.max = rs->nReads + 1; ah.dim = arrayDimAfter (1 << 16, 72, rs->nReads + 1);
  if (fwrite (&ah, sizeof (ah), 1, f) != 1) return 0;
  for (int i = 0 ; i < ah.dim ; ++i)
    { memset (&rec, 0, sizeof (rec));
      if (i >= 1 && i <= rs->nReads)
        { const OrcRead *rd = &rs->reads[i];
          rec.len = rd->len; rec.nHit = rd->nHit; rec.nMiss = rd->nMiss;
          for (int j = 0 ; j < 4 ; ++j) rec.nCopy[j] = rd->nCopy[j];
        }
      if (fwrite (&rec, sizeof (rec), 1, f) != 1) return 0;
    }
  for (int i = 1 ; i <= rs->nReads ; ++i)
    { uint64_t a = rs->hitStart[i], n = rs->hitStart[i + 1] - a;
      if (!n) continue;
      if (fwrite (rs->hit + a, sizeof (uint32_t), n, f) != n) return 0;
      if (fwrite (rs->dx + a, sizeof (uint16_t), n, f) != n) return 0;
    }
  return 1;
}
