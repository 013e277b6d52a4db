/* mg_pack.c — bases (one per byte, values 0..3 in the low two bits: seqio.c:643-652 after the callers' N -> 0 patch)
 * to the 2-bit packed words the kernels read, on the host: 4x fewer bytes cross PCIe than with the device-side pack.
 *
 * Word layout (include/modgpu.h): base i of the stream in bits [30 - 2*(i%16), 32 - 2*(i%16)) of word i/16, i.e. the
 * first base in the most significant bits.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <immintrin.h>
#include "modgpu.h"
#include "mg_knobs.h"

static void packScalar (const unsigned char *b, uint64_t nWords, U32 *w)
{
  for (uint64_t i = 0 ; i < nWords ; ++i, b += 16)
    { U32 x = 0;
      for (int j = 0 ; j < 16 ; ++j) x |= (U32) (b[j] & 3) << (30 - 2 * j);
      w[i] = x;
    }
}

/* 128 bases -> 8 words per step: mask to two bits; pairs of bases 4*a + b (maddubs), pairs of pairs 16*p + q (madd):
 * one output byte per dword lane; the four vectors' lanes packed down to bytes; bytes reversed inside every word
 * (first base in the top byte of a little-endian word); the two 128-bit lanes interleaved back into stream order */
__attribute__ ((target ("avx2")))
static void packAvx2 (const unsigned char *b, uint64_t nWords, U32 *w)
{
  const __m256i m3 = _mm256_set1_epi8 (3), k41 = _mm256_set1_epi16 (0x0104), k161 = _mm256_set1_epi32 (0x00010010);
  const __m256i rev = _mm256_setr_epi8 (3, 2, 1, 0, 7, 6, 5, 4, 11, 10, 9, 8, 15, 14, 13, 12, 3, 2, 1, 0, 7, 6, 5, 4, 11, 10, 9, 8, 15, 14, 13, 12);
  const __m256i order = _mm256_setr_epi32 (0, 4, 1, 5, 2, 6, 3, 7);
  uint64_t i = 0;
  for ( ; i + 8 <= nWords ; i += 8, b += 128)
    { __m256i v[4];
      for (int j = 0 ; j < 4 ; ++j)
        { __m256i x = _mm256_and_si256 (_mm256_loadu_si256 ((const __m256i *) (b + 32 * j)), m3);
          x = _mm256_maddubs_epi16 (x, k41);                     /* 4*b0 + b1 per 16-bit lane (k41 bytes: 4, 1) */
          v[j] = _mm256_madd_epi16 (x, k161);                    /* 16*p0 + p1 per 32-bit lane (k161 words: 16, 1) */
        }
      __m256i p = _mm256_packus_epi16 (_mm256_packus_epi32 (v[0], v[1]), _mm256_packus_epi32 (v[2], v[3]));
      p = _mm256_permutevar8x32_epi32 (_mm256_shuffle_epi8 (p, rev), order);
      _mm256_storeu_si256 ((__m256i *) (w + i), p);
    }
  packScalar (b, nWords - i, w + i);
}

/* whole words [0, nBases/16) and, zero padded, the partial last word: words[0 .. ceil(nBases/16)) */
void mgPackWords (const char *bases, U64 nBases, U32 *words)
{
  const int haveAvx2 = mgKnobs ()->noAvx2 == 1 ? 0 : (__builtin_cpu_supports ("avx2") ? 1 : 0);   /* MODGPU_NO_AVX2=1 (tests): the portable loop */
  const uint64_t full = nBases / 16;
  if (haveAvx2) packAvx2 ((const unsigned char *) bases, full, words);
  else packScalar ((const unsigned char *) bases, full, words);
  const int rest = (int) (nBases - full * 16);
  if (rest)
    { U32 x = 0;
      for (int j = 0 ; j < rest ; ++j) x |= (U32) (bases[full * 16 + j] & 3) << (30 - 2 * j);
      words[full] = x;
    }
}

void mgPackHost (const char *bases, U64 nBases, U32 *words)
{
  mgPackWords (bases, nBases, words);
  const U64 nw = (nBases + 15) / 16;
  for (int j = 0 ; j < MG_PACK_PAD ; ++j) words[nw + j] = 0;
}
