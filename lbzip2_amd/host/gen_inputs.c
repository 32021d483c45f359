/*
 * gen_inputs.c -- deterministic stand-ins for the BASELINE.json corpora (none of enwik8/9,
 * Silesia or the kernel tarball exist on the GPU box and there is no network).  Workload
 * plumbing for bench.py and the parity fixtures, not part of the codec.  Integer arithmetic
 * only (xorshift32 13/17/5), so the same seed gives the same bytes everywhere.
 *
 *   lbzgen_rand   top byte of each draw                                
 *   lbzgen_text   28-symbol word soup of SURVEY.md App. B4             (C1/C2 of round 1)
 *   lbzgen_wiki   enwik-like: XML page wrappers, wiki markup, Zipf-distributed English-shaped
 *                 words, entities, digits, UTF-8 interwiki text (>= 190 distinct bytes) and
 *                 verbatim passage repeats -- byte alphabet, 8-bit sort keys, deep ties (C2)
 *   lbzgen_mixed  text / 64-byte records with counters / random / zeros, 16 MB each     (C3)
 *   lbzgen_tar    tar-like: 512-byte headers, source-like members, zero padding to 512   (C5)
 */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static uint32_t xs32(uint32_t *x) { *x ^= *x << 13; *x ^= *x >> 17; *x ^= *x << 5; return *x; }

void lbzgen_rand(uint8_t *out, size_t n, uint32_t seed)
{
  uint32_t x = seed;
  for (size_t i = 0; i < n; i++) out[i] = (uint8_t)(xs32(&x) >> 24);
}

/* The same sequence in pieces: *state is the generator's state (the seed before the first piece) and is left
   where the next piece goes on -- 10^10 bytes (C4 as BASELINE.json words it) need not be held at once. */
void lbzgen_rand_from(uint8_t *out, size_t n, uint32_t *state)
{
  uint32_t x = *state;
  for (size_t i = 0; i < n; i++) out[i] = (uint8_t)(xs32(&x) >> 24);
  *state = x;
}

void lbzgen_text(uint8_t *out, size_t n, uint32_t seed)
{
  static __thread char words[4096][10];
  static __thread uint8_t wlen[4096];
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

/* ------------------------------------------------------------------ enwik-like text */
#define NWORDS 32768
struct wiki {
  uint32_t x;
  uint8_t *out;
  size_t n, o;
  char (*w)[16];
  uint8_t *wl;
  uint16_t (*ph)[6];          /* collocations: up to 6 word ranks, 0xFFFF-terminated */
};
#define NPHRASES 8192

static const char *const top_words[] = {
  "the", "of", "and", "in", "to", "a", "is", "was", "for", "as", "by", "with", "that", "on", "his", "he",
  "from", "at", "it", "an", "are", "were", "which", "this", "be", "or", "has", "had", "also", "its", "not", "but",
  "first", "one", "their", "have", "new", "after", "who", "they", "two", "her", "she", "been", "other", "when",
  "time", "during", "there", "into", "all", "more", "may", "years", "school", "over", "only", "year", "most",
  "would", "world", "city", "some", "where", "can", "between", "later", "three", "state", "such", "then",
  "national", "used", "made", "known", "under", "many", "university", "united", "while", "part", "season",
  "team", "these", "american", "than", "film", "second", "born", "south", "became", "states", "war", "through",
  "being", "including", "both", "before", "north", "high", "however", "people", "family", "early", "history",
  "album", "area", "them", "series", "against", "until", "since", "district", "county", "name", "work", "life",
  "group", "music", "following", "number", "company", "several", "four", "called", "played", "released", "career",
};
static const char *const onsets[] = { "", "b", "c", "d", "f", "g", "h", "j", "k", "l", "m", "n", "p", "r", "s", "t", "v", "w",
  "st", "tr", "ch", "sh", "th", "pr", "br", "cl", "gr", "pl", "z", "sp", "y", "x" };
static const char *const vowels[] = { "a", "e", "i", "o", "u", "e", "a", "i", "o", "ea", "ou", "io", "ai", "ee", "ie", "y" };
static const char *const codas[] = { "", "", "", "n", "r", "s", "t", "l", "d", "m", "ng", "nt", "st", "rs", "ck", "ll",
  "ss", "nd", "ry", "ly", "ty", "er", "al", "ic", "ed", "es", "on", "ion", "ment", "ing", "ous", "ate" };

static inline void put(struct wiki *g, const char *s, size_t len)
{
  if (len > g->n - g->o) len = g->n - g->o;
  memcpy(g->out + g->o, s, len);
  g->o += len;
}
static inline void puts_(struct wiki *g, const char *s) { put(g, s, strlen(s)); }
static inline void putc_(struct wiki *g, char c) { if (g->o < g->n) g->out[g->o++] = (uint8_t)c; }

/* log-uniform rank: a Zipf(1) law on a dyadic scale */
static inline uint32_t zipf(struct wiki *g)
{
  const uint32_t r = xs32(&g->x);
  const uint32_t L = (r >> 8) % 18u;                     /* 0..15 octaves; 16, 17: anywhere in the vocabulary */
  if (L == 0) return (r >> 16) & 1u;
  if (L >= 16u) return (r >> 13) % NWORDS;
  return ((1u << (L - 1)) + ((r >> 12) & ((1u << (L - 1)) - 1u)) + 1u) % NWORDS;
}
static inline void word(struct wiki *g, int cap)
{
  const uint32_t k = zipf(g);
  if (cap && g->o < g->n) {
    g->out[g->o++] = (uint8_t)(g->w[k][0] - 32);
    put(g, g->w[k] + 1, g->wl[k] - 1u);
  } else {
    put(g, g->w[k], g->wl[k]);
  }
}
static inline void put_rank(struct wiki *g, uint32_t k, int cap)
{
  if (cap && g->o < g->n) {
    g->out[g->o++] = (uint8_t)(g->w[k][0] - 32);
    put(g, g->w[k] + 1, g->wl[k] - 1u);
  } else {
    put(g, g->w[k], g->wl[k]);
  }
}
/* a collocation: phrases recur as wholes, which is where natural text gets its ties of depth 10-40 */
static void phrase(struct wiki *g, int cap)
{
  const uint32_t r = xs32(&g->x);
  const uint32_t L = (r >> 8) % 14u;
  const uint32_t k = L == 0 ? ((r >> 16) & 1u) : ((1u << (L - 1)) + ((r >> 12) & ((1u << (L - 1)) - 1u)) + 1u) % NPHRASES;
  for (uint32_t i = 0; i < 6u && g->ph[k][i] != 0xFFFFu; i++) {
    if (i) putc_(g, ' ');
    put_rank(g, g->ph[k][i], cap && i == 0u);
  }
}
static void number(struct wiki *g, uint32_t lo, uint32_t span)
{
  char b[16];
  const int l = snprintf(b, sizeof b, "%u", lo + xs32(&g->x) % span);
  put(g, b, (size_t)l);
}
static void utf8_word(struct wiki *g, uint32_t script)
{
  /* lead bytes per script; continuation bytes 0x80..0xBF */
  static const uint8_t lead2[] = { 0xC3, 0xC2, 0xC4, 0xC5, 0xCE, 0xCF, 0xD0, 0xD1, 0xD2, 0xD7, 0xD8, 0xD9, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xD3, 0xD4, 0xD5, 0xD6, 0xDA, 0xDB, 0xC8 };
  static const uint8_t lead3[] = { 0xE3, 0xE4, 0xE5, 0xE6, 0xE7, 0xE8, 0xE9, 0xE0, 0xEA, 0xEB, 0xEC, 0xED, 0xE1, 0xEF };
  const uint32_t len = 2u + xs32(&g->x) % 5u;
  for (uint32_t i = 0; i < len; i++) {
    const uint32_t r = xs32(&g->x);
    if (script < 12u) {
      putc_(g, (char)lead2[(2u * script + (r & 1u)) % 24u]);
      putc_(g, (char)(0x80u + ((r >> 4) & 0x3Fu)));
    } else {
      putc_(g, (char)lead3[(script + (r & 3u)) % 14u]);
      putc_(g, (char)(0x80u + ((r >> 4) & 0x3Fu)));
      putc_(g, (char)(0x80u + ((r >> 12) & 0x3Fu)));
    }
  }
}

static void sentence(struct wiki *g)
{
  const uint32_t nw = 6u + xs32(&g->x) % 22u;
  for (uint32_t i = 0; i < nw && g->o < g->n; i++) {
    const uint32_t r = xs32(&g->x);
    const uint32_t k = r % 64u;
    if (i) putc_(g, ' ');
    if (k == 0) {                                        /* [[link]] or [[link|text]] */
      puts_(g, "[[");
      word(g, 1);
      if (r & 0x100u) { putc_(g, ' '); word(g, 0); }
      if (r & 0x200u) { putc_(g, '|'); word(g, 0); }
      puts_(g, "]]");
    } else if (k == 1) {
      puts_(g, "[[");
      number(g, 1000u, 1010u);
      puts_(g, "]]");
    } else if (k == 2) {
      puts_(g, (r & 0x100u) ? "'''" : "''");
      word(g, 0);
      puts_(g, (r & 0x100u) ? "'''" : "''");
    } else if (k == 3) {
      puts_(g, "&quot;");
      word(g, 0);
      putc_(g, ' ');
      word(g, 0);
      puts_(g, "&quot;");
    } else if (k == 4) {
      number(g, 1u, (r & 0x100u) ? 100u : 100000u);
    } else if (k == 5) {
      puts_(g, "&lt;ref&gt;");
      word(g, 1);
      puts_(g, ", p. ");
      number(g, 1u, 900u);
      puts_(g, "&lt;/ref&gt;");
    } else if (k == 6) {
      putc_(g, '(');
      word(g, 0);
      putc_(g, ')');
    } else if (k == 7) {
      word(g, 1);
      puts_(g, "'s");
    } else if (k == 10) {
      number(g, 1u, 100u);
      putc_(g, '%');
    } else if (k == 11) {
      puts_(g, (r & 0x100u) ? "\xe2\x80\x9c" : "\xe2\x80\x98");
      phrase(g, 0);
      puts_(g, (r & 0x100u) ? "\xe2\x80\x9d" : "\xe2\x80\x99");
    } else if (k >= 12 && k < 36) {
      phrase(g, i == 0);
    } else {
      word(g, i == 0 || k == 8 || k == 9);
    }
    if (k >= 56u && k < 61u && i + 1u < nw) putc_(g, ',');
    else if (k == 61u) putc_(g, ';');
    else if (k == 62u) puts_(g, " &amp;");
    else if (k == 63u) puts_(g, " &mdash;");
  }
  {
    const uint32_t e = xs32(&g->x) % 32u;
    puts_(g, e == 0u ? "? " : e == 1u ? "! " : e == 2u ? ".\xe2\x80\x94" : e == 3u ? ": " : ". ");
  }
}

static void template_(struct wiki *g)
{
  static const char *const names[] = { "cite web", "cite book", "Infobox", "main", "see also", "Taxobox", "fact", "cite news" };
  static const char *const keys[] = { "title", "url", "author", "year", "publisher", "name", "image", "caption", "date", "accessdate", "location", "pages" };
  const uint32_t r = xs32(&g->x);
  puts_(g, "{{");
  puts_(g, names[r % 8u]);
  const uint32_t nk = (r >> 8) % 6u;
  for (uint32_t i = 0; i < nk; i++) {
    const uint32_t q = xs32(&g->x);
    puts_(g, (r & 0x80000000u) ? "\n| " : "|");
    puts_(g, keys[q % 12u]);
    puts_(g, (r & 0x80000000u) ? " = " : "=");
    if (q % 12u == 1u) { puts_(g, "http://www."); word(g, 0); puts_(g, (q & 0x1000u) ? ".com/" : ".org/"); word(g, 0); puts_(g, ".html"); }
    else if (q % 12u == 3u) number(g, 1800u, 207u);
    else if (q % 12u == 8u || q % 12u == 9u) { number(g, 2000u, 7u); putc_(g, '-'); number(g, 10u, 3u); putc_(g, '-'); number(g, 10u, 19u); }
    else { word(g, 1); putc_(g, ' '); word(g, 0); }
  }
  puts_(g, (r & 0x80000000u) ? "\n}}\n" : "}}");
}


/* Bot-written gazetteer entries (enwik9 holds tens of thousands): fixed sentences around numbers
   and names -- many copies of moderately long strings, i.e. large groups with deep ties */
static void census(struct wiki *g)
{
  puts_(g, "\'\'\'"); word(g, 1); puts_(g, "\'\'\' is a "); puts_(g, (xs32(&g->x) & 1u) ? "town" : "city");
  puts_(g, " located in [["); word(g, 1); puts_(g, " County, "); word(g, 1); puts_(g, "]]. As of the [[2000]] census, the ");
  puts_(g, "town had a total population of "); number(g, 50u, 90000u);
  puts_(g, ".\n\n== Geography ==\nAccording to the [[United States Census Bureau]], the town has a total area of ");
  number(g, 1u, 300u); putc_(g, '.'); number(g, 0u, 10u); puts_(g, " [[square kilometer|km&sup2;]] ("); number(g, 1u, 100u); putc_(g, '.'); number(g, 0u, 10u);
  puts_(g, " [[square mile|mi&sup2;]]). "); number(g, 1u, 300u); putc_(g, '.'); number(g, 0u, 10u);
  puts_(g, " km&sup2; of it is land and "); number(g, 0u, 9u); putc_(g, '.'); number(g, 0u, 10u);
  puts_(g, " km&sup2; of it is water.\n\n== Demographics ==\nAs of the [[census]] of [[2000]], there are ");
  number(g, 50u, 90000u); puts_(g, " people, "); number(g, 20u, 30000u); puts_(g, " households, and "); number(g, 10u, 20000u);
  puts_(g, " families residing in the town. The [[population density]] is "); number(g, 1u, 900u); putc_(g, '.'); number(g, 0u, 10u);
  puts_(g, "/km&sup2;. There are "); number(g, 20u, 40000u); puts_(g, " housing units at an average density of "); number(g, 1u, 400u); putc_(g, '.'); number(g, 0u, 10u);
  puts_(g, "/km&sup2;. The racial makeup of the town is "); number(g, 40u, 59u); putc_(g, '.'); number(g, 10u, 89u);
  puts_(g, "% [[White (U.S. Census)|White]], "); number(g, 0u, 30u); putc_(g, '.'); number(g, 10u, 89u);
  puts_(g, "% [[African American (U.S. Census)|African American]], and "); number(g, 0u, 5u); putc_(g, '.'); number(g, 10u, 89u);
  puts_(g, "% from two or more races.\n\nThe median income for a household in the town is $"); number(g, 20u, 70u); putc_(g, ','); number(g, 100u, 899u);
  puts_(g, ", and the median income for a family is $"); number(g, 25u, 80u); putc_(g, ','); number(g, 100u, 899u);
  puts_(g, ". The [[per capita income]] for the town is $"); number(g, 10u, 30u); putc_(g, ','); number(g, 100u, 899u); puts_(g, ".\n\n");
}

static void page(struct wiki *g, uint32_t *pid)
{
  static const char *const langs[] = { "de", "fr", "es", "pl", "it", "nl", "el", "el", "ru", "bg", "uk", "he", "ar", "fa",
                                       "ja", "zh", "zh-min-nan", "ko", "ja", "th", "ko", "zh", "ja", "ko" };
  char b[96];
  uint32_t r = xs32(&g->x);
  puts_(g, "  <page>\n    <title>");
  word(g, 1);
  if (r & 1u) { putc_(g, ' '); word(g, (r >> 1) & 1u); }
  if ((r & 12u) == 12u) { putc_(g, ' '); putc_(g, '('); word(g, 0); putc_(g, ')'); }
  puts_(g, "</title>\n    <id>");
  *pid += 1u + (r >> 8) % 7u;
  put(g, b, (size_t)snprintf(b, sizeof b, "%u", *pid));
  puts_(g, "</id>\n    <revision>\n      <id>");
  put(g, b, (size_t)snprintf(b, sizeof b, "%u", 15000000u + xs32(&g->x) % 30000000u));
  puts_(g, "</id>\n      <timestamp>");
  r = xs32(&g->x);
  put(g, b, (size_t)snprintf(b, sizeof b, "200%u-%02u-%02uT%02u:%02u:%02uZ", 2u + r % 5u, 1u + (r >> 3) % 12u, 1u + (r >> 7) % 28u,
                             (r >> 12) % 24u, (r >> 17) % 60u, (r >> 23) % 60u));
  puts_(g, "</timestamp>\n      <contributor>\n");
  r = xs32(&g->x);
  if (r % 4u) {
    puts_(g, "        <username>");
    word(g, 1);
    if (r & 16u) number(g, 1u, 99u);
    puts_(g, "</username>\n        <id>");
    number(g, 100u, 900000u);
    puts_(g, "</id>\n");
  } else {
    puts_(g, "        <ip>");
    put(g, b, (size_t)snprintf(b, sizeof b, "%u.%u.%u.%u", 1u + (r >> 4) % 220u, (r >> 12) % 256u, (r >> 20) % 256u, r >> 24));
    puts_(g, "</ip>\n");
  }
  puts_(g, "      </contributor>\n");
  if (r & 0x100u) puts_(g, "      <minor />\n");
  if (r & 0x600u) { puts_(g, "      <comment>"); word(g, 0); putc_(g, ' '); word(g, 0); if (r & 0x800u) { puts_(g, " [[WP:"); word(g, 1); puts_(g, "]]"); } puts_(g, "</comment>\n"); }
  puts_(g, "      <text xml:space=\"preserve\">");
  r = xs32(&g->x);
  if (r % 16u == 1u) {
    census(g);
  } else if (r % 8u == 0u) {                              /* redirect stub */
    puts_(g, "#REDIRECT [[");
    word(g, 1); putc_(g, ' '); word(g, 0);
    puts_(g, "]]");
  } else {
    const size_t start = g->o;
    const uint32_t npar = 2u + (r >> 4) % 14u;
    if (r & 0x100000u) template_(g);
    for (uint32_t p = 0; p < npar && g->o < g->n; p++) {
      const uint32_t q = xs32(&g->x);
      if (q % 5u == 0u && p) {
        const char *eq = (q & 0x100u) ? "===" : "==";
        puts_(g, eq); putc_(g, ' '); word(g, 1); if (q & 0x200u) { putc_(g, ' '); word(g, 0); } putc_(g, ' '); puts_(g, eq); putc_(g, '\n');
      }
      if (q % 20u == 7u && g->o > 200000u) {
        /* verbatim repeat of an earlier passage (quotations, transcluded boilerplate, mirrored
           pages): mostly short, a few long */
        const uint32_t q2 = xs32(&g->x);
        const size_t back = 1000u + (size_t)(q2 % 600000u) % (g->o - 1000u);
        size_t len = (q2 >> 20) % 16u < 9u ? 40u + (q2 >> 8) % 200u : ((q2 >> 20) % 16u < 15u ? 200u + (q2 >> 8) % 1000u : 1000u + (q2 >> 8) % 4000u);
        if (len > back) len = back;
        const size_t s = g->o - back;
        if (len > g->n - g->o) len = g->n - g->o;
        memmove(g->out + g->o, g->out + s, len);
        g->o += len;
        putc_(g, '\n');
        continue;
      }
      if (q % 16u == 3u) {                                 /* bullet list */
        const uint32_t ni = 2u + (q >> 8) % 8u;
        for (uint32_t i = 0; i < ni; i++) {
          puts_(g, (q & 0x10000u) ? "* [[" : "*[[");
          word(g, 1); putc_(g, ' '); word(g, 0);
          puts_(g, "]]");
          if (q & 0x20000u) { puts_(g, " - "); word(g, 0); putc_(g, ' '); word(g, 0); }
          putc_(g, '\n');
        }
      } else if (q % 16u == 11u) {                         /* table */
        const uint32_t nr = 2u + (q >> 8) % 10u;
        puts_(g, "{| class=\"wikitable\"\n");
        for (uint32_t i = 0; i < nr; i++) {
          puts_(g, "|-\n| "); word(g, 1); puts_(g, " || "); number(g, 1u, 5000u); puts_(g, " || "); number(g, 1900u, 107u); putc_(g, '\n');
        }
        puts_(g, "|}\n");
      } else {
        const uint32_t ns = 1u + (q >> 8) % 7u;
        for (uint32_t i = 0; i < ns; i++) sentence(g);
        if ((q >> 12) % 8u == 0u) template_(g);
      }
      puts_(g, "\n\n");
    }
    r = xs32(&g->x);
    if (r & 1u) { puts_(g, "[[Category:"); word(g, 1); putc_(g, ' '); word(g, 0); puts_(g, "]]\n"); }
    if (r & 2u) { puts_(g, "[[Category:"); number(g, 1700u, 300u); putc_(g, ' '); word(g, 0); puts_(g, "]]\n"); }
    if ((r & 12u) == 0u && g->o - start > 2000u) {
      const uint32_t nl = 3u + (r >> 8) % 14u;             /* interwiki links: UTF-8 in a dozen scripts */
      uint32_t l = (r >> 16) % 24u;
      for (uint32_t i = 0; i < nl; i++) {
        puts_(g, "[["); puts_(g, langs[l]); putc_(g, ':');
        utf8_word(g, l);
        if (xs32(&g->x) & 1u) { putc_(g, ' '); utf8_word(g, l); }
        puts_(g, "]]\n");
        l = (l + 1u + xs32(&g->x) % 3u) % 24u;
      }
    }
  }
  puts_(g, "</text>\n    </revision>\n  </page>\n");
}

void lbzgen_wiki(uint8_t *out, size_t n, uint32_t seed)
{
  static __thread char w[NWORDS][16];
  static __thread uint8_t wl[NWORDS];
  static __thread uint16_t ph[NPHRASES][6];
  struct wiki g = { seed ? seed : 1u, out, n, 0, w, wl, ph };
  /* the vocabulary is the same for every seed (a language), the text is not */
  uint32_t vx = 0x9E3779B9u;
  const uint32_t ntop = sizeof top_words / sizeof *top_words;
  for (uint32_t k = 0; k < NWORDS; k++) {
    if (k < ntop) {
      wl[k] = (uint8_t)strlen(top_words[k]);
      memcpy(w[k], top_words[k], wl[k]);
      continue;
    }
    const uint32_t nsyl = k < 512u ? 1u + xs32(&vx) % 2u : 1u + xs32(&vx) % 4u;
    uint32_t l = 0;
    for (uint32_t s = 0; s < nsyl && l < 9u; s++) {
      const uint32_t r = xs32(&vx);
      const char *a = onsets[r % 32u], *b = vowels[(r >> 8) % 16u], *c = (s + 1u == nsyl) ? codas[(r >> 16) % 32u] : codas[(r >> 16) % 8u];
      for (; *a && l < 15u; a++) w[k][l++] = *a;
      for (; *b && l < 15u; b++) w[k][l++] = *b;
      for (; *c && l < 15u; c++) w[k][l++] = *c;
    }
    if (l < 2u) w[k][l++] = 'e';
    wl[k] = (uint8_t)l;
  }
  {
    struct wiki v = g;
    v.x = 0x2545F491u;
    for (uint32_t k = 0; k < NPHRASES; k++) {
      const uint32_t len = 2u + xs32(&v.x) % 5u;
      for (uint32_t i = 0; i < 6u; i++) ph[k][i] = i < len ? (uint16_t)zipf(&v) : 0xFFFFu;
    }
  }
  uint32_t pid = 10u + seed % 1000u;
  if (n) puts_(&g, "<mediawiki xmlns=\"http://www.mediawiki.org/xml/export-0.3/\" xml:lang=\"en\">\n");
  while (g.o < n) page(&g, &pid);
}

/* ------------------------------------------------------------------ C3: mixed entropy */
void lbzgen_mixed(uint8_t *out, size_t n, uint32_t seed)
{
  const size_t SEG = 16u << 20;
  uint32_t x = seed ? seed : 1u;
  size_t o = 0;
  for (uint32_t s = 0; o < n; s++) {
    const size_t len = n - o < SEG ? n - o : SEG;
    switch (s % 5u) {
      case 0: lbzgen_wiki(out + o, len, seed + s); break;
      case 1: {                                            /* fixed 64-byte records with counters */
        for (size_t i = 0; i < len; i++) {
          const size_t rec = (o + i) / 64u, f = (o + i) % 64u;
          uint8_t b;
          if (f < 4u) b = (uint8_t)(rec >> (8u * f));                  /* little-endian counter */
          else if (f < 8u) b = (uint8_t)((rec * 2654435761u) >> (8u * (f - 4u)));
          else if (f < 16u) b = (uint8_t)("RECORD\0\1"[f - 8u]);
          else if (f < 24u) b = (uint8_t)((rec / 97u) >> (8u * ((f - 16u) & 3u)));
          else if (f < 56u) b = (uint8_t)(((rec >> 3) + f) & 0x1Fu);
          else b = 0;
          out[o + i] = b;
        }
        break;
      }
      case 2: for (size_t i = 0; i < len; i++) out[o + i] = (uint8_t)(xs32(&x) >> 24); break;
      case 3: lbzgen_text(out + o, len, seed + s); break;
      default: memset(out + o, 0, len); break;
    }
    o += len;
  }
}

/* ------------------------------------------------------------------ C5: tar-like source tree */
void lbzgen_tar(uint8_t *out, size_t n, uint32_t seed)
{
  static const char *const kw[] = { "static", "int", "struct", "return", "if", "else", "for", "while", "unsigned", "long", "const",
    "void", "char", "goto", "break", "case", "switch", "sizeof", "NULL", "err", "ret", "dev", "priv", "flags", "lock", "list",
    "inline", "u32", "u8", "size_t", "mutex_lock", "mutex_unlock", "spin_lock_irqsave", "kfree", "kmalloc", "GFP_KERNEL", "container_of",
    "EXPORT_SYMBOL", "module_init", "pr_err", "dev_err", "EINVAL", "ENOMEM", "unlikely", "likely", "data", "len", "buf", "i", "n" };
  const uint32_t nkw = sizeof kw / sizeof *kw;
  static __thread char idn[2048][14];
  static __thread uint8_t idl[2048];
  uint32_t x = seed ? seed : 1u, vx = 0x51ED270Bu;
  for (uint32_t k = 0; k < 2048u; k++) {
    uint32_t l = 0;
    const uint32_t parts = 1u + xs32(&vx) % 3u;
    for (uint32_t p = 0; p < parts && l < 10u; p++) {
      const uint32_t r = xs32(&vx);
      const char *a = onsets[r % 32u], *b = vowels[(r >> 8) % 16u], *c = codas[(r >> 16) % 32u];
      if (p) idn[k][l++] = '_';
      for (; *a && l < 13u; a++) idn[k][l++] = *a;
      for (; *b && l < 13u; b++) idn[k][l++] = *b;
      for (; *c && l < 13u; c++) idn[k][l++] = *c;
    }
    idl[k] = (uint8_t)l;
  }
  static const char lic[] =
    "// SPDX-License-Identifier: GPL-2.0-only\n/*\n * This program is free software; you can redistribute it and/or modify\n"
    " * it under the terms of the GNU General Public License version 2 as\n * published by the Free Software Foundation.\n */\n\n";
  size_t o = 0;
  uint32_t fileno_ = 0;
#define PUT(s, l) do { size_t l_ = (l); if (l_ > n - o) l_ = n - o; memcpy(out + o, (s), l_); o += l_; } while (0)
#define PUTS(s) PUT((s), strlen(s))
#define IDENT() do { const uint32_t r_ = xs32(&x); const uint32_t k_ = ((r_ % 2048u) * ((r_ >> 11) % 2048u)) >> 11; PUT(idn[k_], idl[k_]); } while (0)
  while (o < n) {
    /* 512-byte ustar-like header */
    uint8_t h[512];
    memset(h, 0, sizeof h);
    const uint32_t r = xs32(&x);
    const uint32_t flen = 600u + ((r % 4096u) * ((r >> 12) % 4096u) >> 6) % 90000u;
    int nl = snprintf((char *)h, 100, "linux-6.1/drivers/%.*s/%.*s_%u.c", idl[r % 64u], idn[r % 64u], idl[(r >> 8) % 2048u], idn[(r >> 8) % 2048u], fileno_++);
    (void)nl;
    snprintf((char *)h + 100, 8, "%07o", 0644u);
    snprintf((char *)h + 108, 8, "%07o", 0u);
    snprintf((char *)h + 116, 8, "%07o", 0u);
    snprintf((char *)h + 124, 12, "%011o", flen);
    snprintf((char *)h + 136, 12, "%011o", 1670000000u + fileno_ * 37u);
    memset(h + 148, ' ', 8);
    h[156] = '0';
    memcpy(h + 257, "ustar\0" "00", 8);
    memcpy(h + 265, "root", 4);
    memcpy(h + 297, "root", 4);
    uint32_t sum = 0;
    for (int i = 0; i < 512; i++) sum += h[i];
    snprintf((char *)h + 148, 8, "%06o", sum);
    PUT(h, 512);
    const size_t end = o + flen < n ? o + flen : n;
    PUT(lic, sizeof lic - 1);
    for (uint32_t i = 0, ni = 2u + xs32(&x) % 8u; i < ni && o < end; i++) {
      PUTS("#include <linux/"); IDENT(); PUTS(".h>\n");
    }
    PUTS("\n");
    while (o < end) {
      const uint32_t q = xs32(&x);
      if (q % 16u == 0u && o > 100000u) {                  /* copy-pasted code: an earlier stretch again */
        const uint32_t q2 = xs32(&x);
        const size_t back = 2000u + (size_t)(q2 % 500000u) % (o - 2000u);
        size_t len = 100u + (q2 >> 12) % 1500u;
        if (len > back) len = back;
        if (len > end - o) len = end - o;
        memmove(out + o, out + o - back, len);
        o += len;
        PUTS("\n");
        continue;
      }
      /* a function */
      PUTS("static "); PUTS(kw[1u + q % 11u]); PUTS(" "); IDENT(); PUTS("(struct "); IDENT(); PUTS(" *"); IDENT(); PUTS(")\n{\n");
      const uint32_t ns = 2u + (q >> 8) % 24u;
      uint32_t depth = 1;
      for (uint32_t s = 0; s < ns && o < end; s++) {
        const uint32_t t = xs32(&x);
        for (uint32_t d = 0; d < depth; d++) PUTS("\t");
        switch (t % 8u) {
          case 0: PUTS("if ("); IDENT(); PUTS((t & 256u) ? " == NULL" : " < 0"); PUTS(") {\n"); depth++; break;
          case 1: if (depth > 1u) { o -= 1u; depth--; PUTS("}\n"); } else { PUTS("return "); IDENT(); PUTS(";\n"); } break;
          case 2: IDENT(); PUTS(" = "); IDENT(); PUTS("("); IDENT(); PUTS(", "); PUTS(kw[t / 8u % nkw]); PUTS(");\n"); break;
          case 3: IDENT(); PUTS("->"); IDENT(); PUTS(" = "); { char b[16]; PUT(b, (size_t)snprintf(b, sizeof b, "0x%x", (t >> 8) & 0xFFFFu)); } PUTS(";\n"); break;
          case 4: PUTS("/* "); IDENT(); PUTS(" "); IDENT(); PUTS(" "); PUTS(kw[t / 8u % nkw]); PUTS(" */\n"); break;
          case 5: PUTS(kw[30u + t / 8u % 14u]); PUTS("(&"); IDENT(); PUTS("->"); IDENT(); PUTS(");\n"); break;
          case 6: PUTS("for (i = 0; i < "); IDENT(); PUTS("; i++)\n"); for (uint32_t d = 0; d <= depth; d++) PUTS("\t"); IDENT(); PUTS("[i] = 0;\n"); break;
          default: PUTS("ret = "); IDENT(); PUTS("("); IDENT(); PUTS(");\n"); break;
        }
      }
      while (depth > 1u) { depth--; for (uint32_t d = 0; d < depth; d++) PUTS("\t"); PUTS("}\n"); }
      PUTS("\treturn 0;\n}\n\n");
    }
    o = end;
    const size_t padto = (o + 511u) & ~(size_t)511u;       /* members are padded with zeros to 512 */
    while (o < padto && o < n) out[o++] = 0;
  }
#undef PUT
#undef PUTS
#undef IDENT
}
