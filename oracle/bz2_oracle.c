/*
 * bz2_oracle.c -- CPU oracle for the bzip2 block-compression hot path.
 *
 * TEST INFRASTRUCTURE (see bz2_oracle.h).  A from-scratch restatement of what the
 * reference computes, stage by stage, written for clarity not speed; algorithms are
 * chosen freely wherever the result is mathematically unique (BWT, sorting, the
 * package-merge lists) and follow the reference's tie-breaking arithmetic exactly
 * where the result is not (Huffman weights, packed cost sums, height search).
 *
 * Pinning: tests/test_oracle_vs_ref.py compares every stage with the compiled
 * reference (oracle/_ref/libref.so) on the reference's own test corpora and on
 * seeded inputs; tests/test_oracle_golden.py checks the committed vectors.
 */
#include "bz2_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* CRC-32, polynomial 0x04C11DB7, MSB first (crctab.c:6-50 is this table;
 * update rule encode.c:103).                                                */
/* ------------------------------------------------------------------------- */
static uint32_t crc_tab[256];
static int crc_ready;

static void
crc_setup(void)
{
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i << 24;
    for (int k = 0; k < 8; k++)
      c = (c & 0x80000000u) ? (c << 1) ^ 0x04C11DB7u : (c << 1);
    crc_tab[i] = c;
  }
  crc_ready = 1;
}

uint32_t
orc_crc32(uint32_t crc, const uint8_t *buf, size_t len)
{
  if (!crc_ready) crc_setup();
  for (size_t i = 0; i < len; i++)
    crc = (crc << 8) ^ crc_tab[(crc >> 24) ^ buf[i]];
  return crc;
}

/* ------------------------------------------------------------------------- */
/* Stage 1: RLE1 + block cut (encode.c:135-336; closing of an open run
 * encode.c:443-447).  Token rules as verified in SURVEY.md App. B2.          */
/* ------------------------------------------------------------------------- */
void
orc_collect(const uint8_t *in, size_t len, uint32_t M, uint8_t *block, orc_collect_t *r)
{
  size_t p = 0;
  uint32_t q = 0;

  memset(r->inuse, 0, 256);
  while (p < len && q < M) {
    uint8_t c = in[p];
    size_t run = 1;
    uint32_t room = M - q;
    while (run < 259 && p + run < len && in[p + run] == c) run++;

    if (run < 4 || room < 4) {
      /* plain copies; a cut inside them is allowed (encode.c:173-189, 201-221) */
      uint32_t k = run < room ? (uint32_t)run : room;
      if (run >= 4 && k > 3) k = 3;
      for (uint32_t i = 0; i < k; i++) block[q++] = c;
      r->inuse[c] = 1;
      p += k;
      if (run >= 4) break;              /* room <= 3 with a long run pending: full */
    } else if (room == 4) {
      /* never leave the 4th byte without room for its count (encode.c:218) */
      block[q++] = c; block[q++] = c; block[q++] = c;
      r->inuse[c] = 1;
      p += 3;
      break;
    } else {
      block[q++] = c; block[q++] = c; block[q++] = c; block[q++] = c;
      block[q++] = (uint8_t)(run - 4);
      r->inuse[c] = 1;
      r->inuse[run - 4] = 1;            /* count bytes are symbols too (encode.c:255,271,446) */
      p += run;
    }
  }
  r->nblock = q;
  r->consumed = p;
  r->crc = orc_crc32(0xFFFFFFFFu, in, p);
}

/* ------------------------------------------------------------------------- */
/* Stage 2: BWT of the cyclic rotations (divbwt.c:1706-1726 defines the
 * result; the algorithm here is prefix doubling, unrelated to divsufsort).   */
/* ------------------------------------------------------------------------- */
static int
cmp_u64(const void *a, const void *b)
{
  uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return x < y ? -1 : x > y;
}

int
orc_is_periodic(const uint8_t *T, int32_t n)
{
  /* smallest period p | n with T[i] == T[i+p] cyclically, via failure function */
  int32_t *f, k = 0, p;
  int res;
  if (n < 2) return 0;
  f = malloc((size_t)n * sizeof *f);
  f[0] = 0;
  for (int32_t i = 1; i < n; i++) {
    while (k > 0 && T[i] != T[k]) k = f[k - 1];
    if (T[i] == T[k]) k++;
    f[i] = k;
  }
  p = n - f[n - 1];
  res = (p < n && n % p == 0);
  free(f);
  return res;
}

int32_t
orc_bwt(const uint8_t *T, int32_t n, uint8_t *bwt)
{
  int32_t *sa, *rank, *rank2, idx;
  uint64_t *pairs;
  uint32_t *cnt;

  if (n == 1) { bwt[0] = T[0]; return 0; }           /* divbwt.c:1712 */

  sa = malloc((size_t)n * sizeof *sa);
  rank = malloc((size_t)n * sizeof *rank);
  rank2 = malloc((size_t)n * sizeof *rank2);
  pairs = malloc((size_t)n * sizeof *pairs);
  cnt = calloc(65537, sizeof *cnt);

  /* depth-2 counting sort; rank = index of the group's first row */
  for (int32_t i = 0; i < n; i++)
    cnt[(((uint32_t)T[i] << 8) | T[i + 1 < n ? i + 1 : 0]) + 1]++;
  for (int i = 0; i < 65536; i++) cnt[i + 1] += cnt[i];
  for (int32_t i = 0; i < n; i++) {
    uint32_t k = ((uint32_t)T[i] << 8) | T[i + 1 < n ? i + 1 : 0];
    rank[i] = (int32_t)cnt[k];
  }
  for (int32_t i = 0; i < n; i++) {
    uint32_t k = ((uint32_t)T[i] << 8) | T[i + 1 < n ? i + 1 : 0];
    sa[cnt[k]++] = i;
  }

  for (int64_t h = 2; h < n; h *= 2) {
    int unsorted = 0;
    memcpy(rank2, rank, (size_t)n * sizeof *rank);
    for (int32_t a = 0; a < n;) {
      int32_t b = a + 1;
      while (b < n && rank[sa[b]] == a) b++;
      if (b - a > 1) {
        int32_t head = a;
        for (int32_t j = a; j < b; j++) {
          int64_t t = sa[j] + h;
          if (t >= n) t -= n;
          pairs[j] = ((uint64_t)(uint32_t)rank[t] << 32) | (uint32_t)sa[j];
        }
        qsort(pairs + a, (size_t)(b - a), sizeof *pairs, cmp_u64);
        for (int32_t j = a; j < b; j++) {
          if (j > a && (pairs[j] >> 32) != (pairs[j - 1] >> 32)) head = j;
          sa[j] = (int32_t)(uint32_t)pairs[j];
          rank2[sa[j]] = head;
          if (j > a && head != j) unsorted = 1;
        }
      }
      a = b;
    }
    { int32_t *t = rank; rank = rank2; rank2 = t; }
    if (!unsorted) break;
  }

  idx = rank[0];                     /* unique unless T = u^k: then the smallest equal row */
  for (int32_t j = 0; j < n; j++)
    bwt[j] = T[sa[j] ? sa[j] - 1 : n - 1];

  free(sa); free(rank); free(rank2); free(pairs); free(cnt);
  return idx;
}

/* ------------------------------------------------------------------------- */
/* Stage 3: MTF + RUNA/RUNB + histogram (encode.c:340-355 map, 360-425 MTF)  */
/* ------------------------------------------------------------------------- */
uint32_t
orc_mtf(const uint8_t *bwt, int32_t n, const uint8_t inuse[256],
        uint16_t *mtfv, uint32_t freq[ORC_MAX_ALPHA + 1], uint32_t *alpha)
{
  uint8_t dense[256], list[256];
  uint32_t ninuse = 0, eob, nm = 0, zeros = 0;

  for (int i = 0; i < 256; i++) { dense[i] = (uint8_t)ninuse; ninuse += inuse[i] != 0; }
  eob = ninuse + 1;
  for (uint32_t i = 0; i <= eob; i++) freq[i] = 0;
  for (int i = 0; i < 256; i++) list[i] = (uint8_t)i;

#define FLUSH_ZEROS()                                                       \
  while (zeros) { uint32_t d = (zeros - 1) & 1;  /* bijective base 2 */     \
                  mtfv[nm++] = (uint16_t)d; freq[d]++; zeros = (zeros - 1) >> 1; }

  for (int32_t i = 0; i < n; i++) {
    uint8_t c = dense[bwt[i]];
    uint32_t pos = 0;
    if (list[0] == c) { zeros++; continue; }
    FLUSH_ZEROS();
    while (list[pos] != c) pos++;
    memmove(list + 1, list, pos);
    list[0] = c;
    mtfv[nm++] = (uint16_t)(pos + 1);
    freq[pos + 1]++;
  }
  FLUSH_ZEROS();
#undef FLUSH_ZEROS
  mtfv[nm++] = (uint16_t)eob;
  freq[eob]++;
  *alpha = eob + 1;
  return nm;
}

/* ------------------------------------------------------------------------- */
/* Stage 4: prefix codes                                                      */
/* ------------------------------------------------------------------------- */

/* leaf weight layout of encode.c:732-741: freq<<32 | depth<<24 | count<<16 | (258-sym) */
static uint64_t
leaf_weight(uint32_t f, uint32_t sym)
{
  return ((uint64_t)f << 32) | 0x10000u | (ORC_MAX_ALPHA - sym);
}

static void
sort_desc(uint64_t *w, uint32_t n)       /* keys are unique: any sort (encode.c:553-567) */
{
  for (uint32_t i = 1; i < n; i++) {
    uint64_t t = w[i];
    uint32_t j = i;
    while (j > 0 && w[j - 1] < t) { w[j] = w[j - 1]; j--; }
    w[j] = t;
  }
}

/* Unrestricted Huffman lengths for the EM M-step (encode.c:713-766 with
 * build_tree :574-615 and compute_depths :619-649).  Depth may reach 30.    */
static void
huffman_lengths(uint8_t *length, const uint32_t *freq, uint32_t as)
{
  uint64_t w[ORC_MAX_ALPHA];
  uint32_t parent[ORC_MAX_ALPHA], depth[ORC_MAX_ALPHA];
  uint32_t internal_at[32], leaves_at[32];
  uint32_t leaf, node, t, i, d;

  for (i = 0; i < as; i++) w[i] = leaf_weight(freq[i] ? freq[i] : 1, i);
  sort_desc(w, as);

  /* Two-queue merge.  Leaves wait in w[0..leaf) (lightest at leaf-1); finished
   * internal nodes wait in w(t..node) (oldest = lightest at node-1).  Node t
   * reuses slot t and keeps that slot's low 16 bits (the symbol id).         */
  leaf = as; node = as;
  for (t = as - 1; t > 0; t--) {
    uint32_t n_int = node - 1 - t;           /* internal nodes available */
    uint64_t a, b;
    if (leaf == 0 || (n_int >= 2 && w[node - 2] < w[leaf - 1])) {
      a = w[node - 1]; b = w[node - 2];
      parent[node - 1] = t; parent[node - 2] = t; node -= 2;
    } else if (n_int == 0 || (leaf >= 2 && w[leaf - 2] <= w[node - 1])) {
      a = w[leaf - 1]; b = w[leaf - 2]; leaf -= 2;
    } else {
      a = w[node - 1]; b = w[leaf - 1];
      parent[node - 1] = t; node -= 1; leaf -= 1;
    }
    {
      uint64_t da = a & 0xFF000000u, db = b & 0xFF000000u;
      w[t] = (w[t] & 0xFFFFu) + ((a + b) & ~(uint64_t)0xFF00FFFFu)
           + (da > db ? da : db) + 0x01000000u;          /* encode.c:609-610 */
    }
  }

  /* internal nodes 1..as-1; node 1 is the root.  Leaves per depth follow
   * from the internal-node census of the level above.                       */
  memset(internal_at, 0, sizeof internal_at);
  depth[1] = 0; internal_at[0] = 1;
  for (i = 2; i < as; i++) { depth[i] = depth[parent[i]] + 1; internal_at[depth[i]]++; }
  leaves_at[0] = 0;
  for (d = 1; d <= 30; d++) leaves_at[d] = 2 * internal_at[d - 1] - internal_at[d];

  /* hand out lengths by rank: heaviest symbols get the shortest (encode.c:750-763) */
  i = 0;
  for (d = 1; d <= 30; d++)
    for (uint32_t k = leaves_at[d]; k > 0; k--, i++)
      length[ORC_MAX_ALPHA - (uint32_t)(w[i] & 0xFFFFu)] = (uint8_t)d;
}

/* Length-limited (<=20) code for one table + canonical codes; returns the bit
 * cost of sending the table and all its symbols (encode.c:882-987).  The
 * reference evaluates a lazy boundary package-merge (encode.c:660-710); the
 * item lists it walks are the textbook package-merge lists, built here
 * explicitly.  Tie rule (from the weight layout): on equal frequency a leaf
 * precedes a package.                                                        */
static uint32_t
limited_code(uint32_t *code, uint8_t *length, const uint32_t *freq, uint32_t as)
{
  enum { LV = 20, MAXI = 2 * ORC_MAX_ALPHA };
  static _Thread_local uint64_t item_f[LV + 1][MAXI];
  static _Thread_local uint16_t leaves_in[LV + 1][MAXI + 1];  /* leaves among first k items */
  uint32_t nitems[LV + 1];
  uint64_t w[ORC_MAX_ALPHA], lf[ORC_MAX_ALPHA];
  uint32_t taken[LV + 2][LV + 2];
  uint32_t want = 2 * as - 2;
  uint32_t best_cost = 0xFFFFFFFFu, best_h = LV;
  uint32_t base[LV + 2];

  for (uint32_t i = 0; i < as; i++) w[i] = leaf_weight(freq[i], i);
  sort_desc(w, as);
  for (uint32_t k = 0; k < as; k++) lf[k] = w[as - 1 - k] >> 32;   /* ascending */

  for (uint32_t lv = 1; lv <= LV; lv++) {
    uint32_t li = 0, pi = 0, k = 0;
    uint32_t npk = lv > 1 ? nitems[lv - 1] / 2 : 0;
    leaves_in[lv][0] = 0;
    while (k < want && (li < as || pi < npk)) {
      uint64_t pf = 0;
      int take_pkg = 0;
      if (pi < npk) {
        pf = item_f[lv - 1][2 * pi] + item_f[lv - 1][2 * pi + 1];
        take_pkg = (li >= as) || (pf < lf[li]);
      }
      if (take_pkg) { item_f[lv][k] = pf; pi++; leaves_in[lv][k + 1] = leaves_in[lv][k]; }
      else { item_f[lv][k] = lf[li++]; leaves_in[lv][k + 1] = leaves_in[lv][k] + 1; }
      k++;
    }
    nitems[lv] = k;
  }

  /* taken[h][d] = leaves used on level h-d when the limit is h */
  memset(taken, 0, sizeof taken);
  for (uint32_t h = 1; h <= LV; h++) {
    uint32_t k = want < nitems[h] ? want : nitems[h];
    for (uint32_t d = 0; d < h; d++) {
      uint32_t lv = h - d, nl;
      if (k > nitems[lv]) k = nitems[lv];
      nl = leaves_in[lv][k];
      taken[h][d] = nl;
      k = 2 * (k - nl);
    }
  }

  for (uint32_t h = 2; h <= LV; h++) {             /* height search, encode.c:913-945 */
    uint32_t cost = 0, rank = 0;
    if ((1ul << h) < as) continue;
    if (taken[h][h - 1] == 0) break;
    for (uint32_t d = 1; d <= h; d++)
      for (uint32_t k = taken[h][d - 1] - taken[h][d]; k > 0; k--, rank++) {
        length[ORC_MAX_ALPHA - (uint32_t)(w[rank] & 0xFFFFu)] = (uint8_t)d;
        cost += (uint32_t)(w[rank] >> 32) * d;
      }
    for (uint32_t v = 1; v < as; v++) {
      int dl = (int)length[v] - (int)length[v - 1];
      cost += 2 * (uint32_t)(dl < 0 ? -dl : dl);
    }
    cost += 5 + as;
    if (cost < best_cost) { best_cost = cost; best_h = h; }
  }

  {
    uint32_t rank = 0, next = 0;
    for (uint32_t d = 1; d <= best_h; d++) {
      uint32_t k = taken[best_h][d - 1] - taken[best_h][d];
      base[d] = next;
      next = (next + k) << 1;
      for (; k > 0; k--, rank++)
        length[ORC_MAX_ALPHA - (uint32_t)(w[rank] & 0xFFFFu)] = (uint8_t)d;
    }
  }
  for (uint32_t v = 0; v < as; v++) code[v] = base[length[v]]++;
  return best_cost;
}

/* Initial partition of the alphabet into nt frequency classes (encode.c:779-841):
 * length 0 inside a table's class, 1 everywhere else.                        */
static void
seed_tables(uint8_t length[ORC_MAX_TREES][ORC_MAX_ALPHA + 1], const uint32_t *freq,
            uint32_t as, uint32_t nm, uint32_t nt)
{
  uint32_t live = 0, a = 0;
  memset(length, 1, ORC_MAX_TREES * (ORC_MAX_ALPHA + 1));
  for (uint32_t v = 0; v < as; v++) live += freq[v] != 0;
  if (nt > live) nt = live;
  for (uint32_t t = 0; nt > 0; t++, nt--) {
    uint32_t f = freq[a], cum = f, b = a + 1;
    live -= f != 0;
    while (live > nt - 1 && cum * nt < nm) {
      f = freq[b++]; cum += f; live -= f != 0;
    }
    if (cum > f && (2 * cum - f) * nt > 2 * nm) {
      cum -= f; live += f != 0; b--;
    }
    memset(&length[t][a], 0, b - a);
    a = b;
    nm -= cum;
  }
}

void
orc_prefix_code(uint16_t *mtfv, uint32_t nm, const uint32_t *freq,
                unsigned cluster_factor, orc_code_t *pc)
{
  static _Thread_local uint32_t tf[ORC_MAX_TREES][ORC_MAX_ALPHA + 1];
  uint32_t as = (uint32_t)mtfv[nm - 1] + 1;
  uint32_t ns = (nm + ORC_GROUP - 1) / ORC_GROUP;
  uint32_t nt = nm > 2400 ? 6 : nm > 1200 ? 5 : nm > 600 ? 4 : nm > 300 ? 3 : nm > 150 ? 2 : 1;
  uint32_t cost = 0, used = 0, seen = 0;

  pc->num_selectors = ns;
  for (uint32_t i = nm; i < ns * ORC_GROUP; i++) mtfv[i] = (uint16_t)as;  /* pad, encode.c:1034 */
  seed_tables(pc->length, freq, as, nm, nt);

  for (unsigned it = 0; it < cluster_factor; it++) {
    uint64_t pack[ORC_MAX_ALPHA + 1];
    /* six 10-bit fields in one word; sums may carry between fields and the
     * reference lets them (encode.c:1050-1061, 858-872)                      */
    for (uint32_t v = 0; v < as; v++) {
      uint64_t x = 0;
      for (int t = ORC_MAX_TREES - 1; t >= 0; t--) x = (x << 10) + pc->length[t][v];
      pack[v] = x;
    }
    pack[as] = 0;
    memset(tf, 0, sizeof tf);
    for (uint32_t g = 0; g < ns; g++) {
      const uint16_t *gs = mtfv + g * ORC_GROUP;
      uint64_t sum = 0;
      uint32_t bt = 0, bc;
      for (int i = 0; i < ORC_GROUP; i++) sum += pack[gs[i]];
      bc = (uint32_t)(sum & 0x3ff);
      for (uint32_t t = 1; t < nt; t++) {
        uint32_t c;
        sum >>= 10;
        c = (uint32_t)(sum & 0x3ff);
        if (c < bc) { bc = c; bt = t; }        /* first minimum wins */
      }
      pc->selector[g] = (uint8_t)bt;
      for (int i = 0; i < ORC_GROUP; i++) tf[bt][gs[i]]++;
    }
    for (uint32_t t = 0; t < nt; t++) huffman_lengths(pc->length[t], tf[t], as);
  }

  /* renumber tables by first use, drop unused ones (encode.c:1088-1111) */
  for (uint32_t g = 0; g < ns && seen != (1u << nt) - 1; g++) {
    uint32_t t = pc->selector[g];
    if (seen & (1u << t)) continue;
    seen |= 1u << t;
    pc->old2new[t] = used;
    pc->new2old[used] = t;
    used++;
    cost += limited_code(pc->code[t], pc->length[t], tf[t], as);
    pc->code[t][as] = 0;
    pc->length[t][as] = 0;
  }

  if (used == 1) {                               /* dummy 2nd table, encode.c:1117-1132 */
    uint32_t t = pc->new2old[0] ^ 1, lg = 0, v;
    while ((2u << lg) <= as) lg++;               /* floor(log2(as)) */
    pc->old2new[t] = 1;
    pc->new2old[1] = t;
    for (v = 0; v < (2u << lg) - as; v++) pc->length[t][v] = (uint8_t)lg;
    if (v < as) cost += 2;
    for (; v < as; v++) pc->length[t][v] = (uint8_t)(lg + 1);
    cost += as + 5;
    used = 2;
  }
  pc->num_trees = used;
  pc->cost = cost;
}

/* ------------------------------------------------------------------------- */
/* Stage 5: exact size (encode.c:460-545) and bit packing (encode.c:1152-1281) */
/* ------------------------------------------------------------------------- */
void
orc_encode_block(const uint8_t *block, const orc_collect_t *c,
                 unsigned cluster_factor, uint16_t *mtfv, orc_block_t *b)
{
  uint8_t *bwt = malloc(c->nblock);
  uint32_t freq[ORC_MAX_ALPHA + 1];
  uint32_t bits, pad;
  uint8_t mtf[ORC_MAX_TREES] = { 0, 1, 2, 3, 4, 5 };

  b->nblock = c->nblock;
  b->crc = c->crc;
  memcpy(b->inuse, c->inuse, 256);
  b->bwt_idx = (uint32_t)orc_bwt(block, (int32_t)c->nblock, bwt);
  b->nmtf = orc_mtf(bwt, (int32_t)c->nblock, c->inuse, mtfv, freq, &b->alpha);
  free(bwt);
  orc_prefix_code(mtfv, b->nmtf, freq, cluster_factor, &b->pc);

  bits = 48 + 32 + 1 + 24 + 3 + 15 + b->pc.cost;
  for (uint32_t g = 0; g < b->pc.num_selectors; g++) {     /* selector MTF, encode.c:482-512 */
    uint8_t t = (uint8_t)b->pc.old2new[b->pc.selector[g]];
    uint32_t j = 0;
    while (mtf[j] != t) j++;
    memmove(mtf + 1, mtf, j);
    mtf[0] = t;
    b->selector_mtf[g] = (uint8_t)j;
    bits += j + 1;
  }
  pad = (8 - (bits & 7)) & 7;                               /* encode.c:514-525 */
  bits += pad;
  b->tree_pad = pad >> 1;
  b->num_selectors_tx = b->pc.num_selectors + (pad & 1);
  if (pad & 1) b->selector_mtf[b->pc.num_selectors] = 0;
  bits += 16;
  for (int i = 0; i < 16; i++) {
    int any = 0;
    for (int j = 0; j < 16; j++) any |= c->inuse[16 * i + j];
    if (any) bits += 16;
  }
  b->out_len = bits >> 3;
}

typedef struct { uint8_t *p; uint64_t acc; unsigned n; } bitw_t;

static void
put(bitw_t *w, unsigned nbits, uint32_t v)
{
  w->acc = (w->acc << nbits) | v;
  w->n += nbits;
  while (w->n >= 8) { w->n -= 8; *w->p++ = (uint8_t)(w->acc >> w->n); }
}

void
orc_transmit(const orc_block_t *b, const uint16_t *mtfv, uint8_t *out)
{
  bitw_t w = { out, 0, 0 };
  uint32_t as = b->alpha, big = 0;
  uint32_t pack[16];

  put(&w, 24, 0x314159); put(&w, 24, 0x265359);
  put(&w, 16, (~b->crc) >> 16); put(&w, 16, (~b->crc) & 0xFFFF);
  put(&w, 1, 0);
  put(&w, 24, b->bwt_idx);

  for (int i = 0; i < 16; i++) {
    pack[i] = 0;
    for (int j = 0; j < 16; j++) pack[i] = (pack[i] << 1) | (b->inuse[16 * i + j] != 0);
    big = (big << 1) | (pack[i] != 0);
  }
  put(&w, 16, big);
  for (int i = 0; i < 16; i++) if (pack[i]) put(&w, 16, pack[i]);

  put(&w, 3, b->pc.num_trees);
  put(&w, 15, b->num_selectors_tx);
  for (uint32_t g = 0; g < b->num_selectors_tx; g++) {
    unsigned v = 1u + b->selector_mtf[g];
    put(&w, v, (1u << v) - 2);
  }

  for (uint32_t t = 0; t < b->pc.num_trees; t++) {
    const uint8_t *len = b->pc.length[b->pc.new2old[t]];
    int a = len[0];
    if (t == 0) a += (a < 4) ? (int)b->tree_pad : -(int)b->tree_pad;   /* encode.c:1235-1241 */
    put(&w, 5, (uint32_t)a);
    for (uint32_t v = 0; v < as; v++) {
      while (a < len[v]) { put(&w, 2, 2); a++; }
      while (a > len[v]) { put(&w, 2, 3); a--; }
      put(&w, 1, 0);
    }
  }

  for (uint32_t g = 0; g < b->pc.num_selectors; g++) {
    uint32_t t = b->pc.selector[g];
    for (int i = 0; i < ORC_GROUP; i++) {
      uint16_t mv = mtfv[g * ORC_GROUP + i];
      if (b->pc.length[t][mv]) put(&w, b->pc.length[t][mv], b->pc.code[t][mv]);
    }
  }
}

/* ------------------------------------------------------------------------- */
/* Whole stream (compress.c:73-118 work units, :246-247 CRC fold, :291-321)   */
/* ------------------------------------------------------------------------- */
size_t
orc_compress_stream(const uint8_t *in, size_t len, unsigned bs100k,
                    uint8_t *out, size_t cap, uint32_t *nblocks)
{
  uint32_t M = bs100k * 100000u, combined = 0, nb = 0;
  uint8_t *block = malloc(M);
  uint16_t *mtfv = malloc(((size_t)M + 1 + ORC_GROUP) * sizeof *mtfv);
  orc_block_t *b = malloc(sizeof *b);
  size_t o = 0;

  if (cap < 14) { o = 0; goto done; }
  out[o++] = 'B'; out[o++] = 'Z'; out[o++] = 'h'; out[o++] = (uint8_t)('0' + bs100k);
  for (size_t off = 0; off < len; off += M) {
    size_t left = len - off < M ? len - off : M;
    const uint8_t *p = in + off;
    while (left > 0) {
      orc_collect_t c;
      orc_collect(p, left, M, block, &c);
      p += c.consumed; left -= c.consumed;
      orc_encode_block(block, &c, 8, mtfv, b);
      if (o + b->out_len + 10 > cap) { o = 0; goto done; }
      orc_transmit(b, mtfv, out + o);
      o += b->out_len;
      combined = ((combined << 1) | (combined >> 31)) ^ ~c.crc;   /* encode.h:38 */
      nb++;
    }
  }
  out[o++] = 0x17; out[o++] = 0x72; out[o++] = 0x45; out[o++] = 0x38; out[o++] = 0x50; out[o++] = 0x90;
  out[o++] = (uint8_t)(combined >> 24); out[o++] = (uint8_t)(combined >> 16);
  out[o++] = (uint8_t)(combined >> 8);  out[o++] = (uint8_t)combined;
done:
  if (nblocks) *nblocks = nb;
  free(block); free(mtfv); free(b);
  return o;
}

/* -u / --sequential (compress.c:129-198): a block takes input until it is full, whatever the slab boundaries */
size_t
orc_compress_seq(const uint8_t *in, size_t len, unsigned bs100k,
                 uint8_t *out, size_t cap, uint32_t *nblocks)
{
  uint32_t M = bs100k * 100000u, combined = 0, nb = 0;
  uint8_t *block = malloc(M);
  uint16_t *mtfv = malloc(((size_t)M + 1 + ORC_GROUP) * sizeof *mtfv);
  orc_block_t *b = malloc(sizeof *b);
  size_t o = 0, pos = 0;

  if (cap < 14) { o = 0; goto done; }
  out[o++] = 'B'; out[o++] = 'Z'; out[o++] = 'h'; out[o++] = (uint8_t)('0' + bs100k);
  while (pos < len) {
    orc_collect_t c;
    orc_collect(in + pos, len - pos, M, block, &c);
    pos += c.consumed;
    orc_encode_block(block, &c, 8, mtfv, b);
    if (o + b->out_len + 10 > cap) { o = 0; goto done; }
    orc_transmit(b, mtfv, out + o);
    o += b->out_len;
    combined = ((combined << 1) | (combined >> 31)) ^ ~c.crc;
    nb++;
  }
  out[o++] = 0x17; out[o++] = 0x72; out[o++] = 0x45; out[o++] = 0x38; out[o++] = 0x50; out[o++] = 0x90;
  out[o++] = (uint8_t)(combined >> 24); out[o++] = (uint8_t)(combined >> 16);
  out[o++] = (uint8_t)(combined >> 8);  out[o++] = (uint8_t)combined;
done:
  if (nblocks) *nblocks = nb;
  free(block); free(mtfv); free(b);
  return o;
}

/* ------------------------------------------------------------------------- */
/* Seeded inputs                                          */
/* ------------------------------------------------------------------------- */
static uint32_t
xs32(uint32_t *x)
{
  *x ^= *x << 13; *x ^= *x >> 17; *x ^= *x << 5;
  return *x;
}

void
orc_gen_rand(uint8_t *out, size_t n, uint32_t seed)
{
  uint32_t x = seed;
  for (size_t i = 0; i < n; i++) out[i] = (uint8_t)(xs32(&x) >> 24);
}

void
orc_gen_text(uint8_t *out, size_t n, uint32_t seed)
{
  static _Thread_local char words[4096][10];
  static _Thread_local uint8_t wlen[4096];
  uint32_t x = seed;
  size_t o = 0;
  for (int w = 0; w < 4096; w++) {
    wlen[w] = (uint8_t)(2 + xs32(&x) % 8);
    for (int k = 0; k < wlen[w]; k++) words[w][k] = (char)('a' + xs32(&x) % 26);
  }
  while (o < n) {
    uint32_t k = xs32(&x) % 4096;
    k = (k * k) >> 12;
    for (int i = 0; i < wlen[k] && o < n; i++) out[o++] = (uint8_t)words[k][i];
    if (o < n) out[o++] = (xs32(&x) % 16 == 0) ? '\n' : ' ';
  }
}

/* ------------------------------------------------------------------------- */
/* pthreads driver on the restatement (cpu_mt.h): bench.py's cpu_baseline "port" */
/* ------------------------------------------------------------------------- */
struct orc_mt_state { uint8_t *block; uint16_t *mtfv; orc_block_t *b; };
static struct orc_mt_state *orc_mt_new(size_t mbs)
{
  struct orc_mt_state *s = malloc(sizeof *s);
  s->block = malloc(mbs);
  s->mtfv = malloc((mbs + 1 + ORC_GROUP) * sizeof *s->mtfv);
  s->b = malloc(sizeof *s->b);
  return s;
}
static void orc_mt_free(struct orc_mt_state *s) { free(s->block); free(s->mtfv); free(s->b); free(s); }
typedef struct { uint32_t out_len, crc, bwt_idx, copies, slab, pad_; } cpu_mt_blk;
#define CPU_MT_BLK_DEFINED
static size_t orc_mt_slab(struct orc_mt_state *s, const uint8_t *in, size_t len, size_t mbs, uint8_t *out,
                          cpu_mt_blk *blk, unsigned *nblk)
{
  size_t o = 0, left = len;
  const uint8_t *p = in;
  unsigned nb = 0;
  while (left > 0) {
    orc_collect_t c;
    orc_collect(p, left, (uint32_t)mbs, s->block, &c);
    p += c.consumed; left -= c.consumed;
    orc_encode_block(s->block, &c, 8, s->mtfv, s->b);
    orc_transmit(s->b, s->mtfv, out + o);
    o += s->b->out_len;
    if (nb < 2) { blk[nb].out_len = s->b->out_len; blk[nb].crc = c.crc; blk[nb].bwt_idx = s->b->bwt_idx; blk[nb].copies = orc_is_periodic(s->block, (int32_t)c.nblock) ? 2u : 1u; }
    nb++;
  }
  *nblk = nb;
  return o;
}
#define CPU_MT_NAME orc_compress_mt
#define CPU_MT_WORKER_STATE struct orc_mt_state
#define cpu_mt_state_new orc_mt_new
#define cpu_mt_state_free orc_mt_free
#define cpu_mt_slab orc_mt_slab
#include "cpu_mt.h"
