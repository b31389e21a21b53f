/* TEST INFRASTRUCTURE (CPU oracle side): dbgen's text pool.
 *
 * dbgen (TPC-H tools 2.17+/3.x, text.c; the TPC's generator is NOT part of /root/reference) fills every comment column with a
 * substring of ONE 300 MiB pool of pseudo-English, generated once from the grammar distributions of dists.dss with its own random
 * stream (seed 933588178, one step per drawn word).  This file restates that generator: sentences of noun / verb /
 * prepositional phrases joined exactly as dbgen joins them (a blank after every word, terminators abutting the previous word, the
 * comma of "J, J N" likewise), until the pool holds `size` bytes.  The distributions are handed in as text (oracle/dbgen.py
 * holds them) so that this file is only the state machine.
 *
 * Pinned by tests/test_dbgen_golden.py: the comment strings the reference carries (core/tests/data/tpch_*_small.parquet,
 * core/tests/tpch-csv/{table}.csv, the s_comment / c_comment columns of answers/q2.slt.part and q10.slt.part) are found at the
 * offsets their row's random streams select.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAX_ENTRIES 64
typedef struct {
  int n;
  int total;
  int cum[MAX_ENTRIES];
  const char* word[MAX_ENTRIES];
  int len[MAX_ENTRIES];
} dist_t;

enum { D_GRAMMAR, D_NP, D_VP, D_NOUNS, D_VERBS, D_ADJECTIVES, D_ADVERBS, D_PREPOSITIONS, D_AUXILIARIES, D_TERMINATORS, D_ARTICLES, D_COUNT };
static const char* NAMES[D_COUNT] = {"grammar", "np", "vp", "nouns", "verbs", "adjectives", "adverbs", "prepositions", "auxillaries", "terminators", "articles"};

typedef struct {
  int64_t seed;
  char* out;
  int64_t len;
} gen_t;

static inline int next_int(gen_t* g, int lo, int hi) {   /* rnd.c UnifInt: the same two IEEE operations */
  g->seed = g->seed * 16807 % 2147483647;
  return lo + (int)((double)g->seed / 2147483647.0 * (double)(hi - lo + 1));
}
static inline int pick(gen_t* g, const dist_t* d) {
  const int r = next_int(g, 0, d->total - 1);
  int i = 0;
  while (d->cum[i] <= r) i++;
  return i;
}
static inline void put(gen_t* g, const char* s, int n) {
  memcpy(g->out + g->len, s, (size_t)n);
  g->len += n;
}
static inline void word(gen_t* g, const dist_t* d) {
  const int i = pick(g, d);
  put(g, d->word[i], d->len[i]);
  put(g, " ", 1);
}
static void noun_phrase(gen_t* g, const dist_t* D) {
  const int s = pick(g, &D[D_NP]);
  const char* syn = D[D_NP].word[s];
  for (int i = 0; i < D[D_NP].len[s]; i++) {
    switch (syn[i]) {
      case 'A': word(g, &D[D_ARTICLES]); break;
      case 'J': word(g, &D[D_ADJECTIVES]); break;
      case 'D': word(g, &D[D_ADVERBS]); break;
      case 'N': word(g, &D[D_NOUNS]); break;
      case ',': g->len -= 1; put(g, ", ", 2); break;
      default: break;
    }
  }
}
static void verb_phrase(gen_t* g, const dist_t* D) {
  const int s = pick(g, &D[D_VP]);
  const char* syn = D[D_VP].word[s];
  for (int i = 0; i < D[D_VP].len[s]; i += 2) {
    switch (syn[i]) {
      case 'D': word(g, &D[D_ADVERBS]); break;
      case 'V': word(g, &D[D_VERBS]); break;
      case 'X': word(g, &D[D_AUXILIARIES]); break;
      default: break;
    }
  }
}
static void sentence(gen_t* g, const dist_t* D) {
  const int s = pick(g, &D[D_GRAMMAR]);
  const char* syn = D[D_GRAMMAR].word[s];
  for (int i = 0; i < D[D_GRAMMAR].len[s]; i += 2) {
    switch (syn[i]) {
      case 'V': verb_phrase(g, D); break;
      case 'N': noun_phrase(g, D); break;
      case 'P': {
        const int p = pick(g, &D[D_PREPOSITIONS]);
        put(g, D[D_PREPOSITIONS].word[p], D[D_PREPOSITIONS].len[p]);
        put(g, " the ", 5);
        noun_phrase(g, D);
        break;
      }
      case 'T': {
        g->len -= 1; /* terminators abut the previous word */
        const int t = pick(g, &D[D_TERMINATORS]);
        put(g, D[D_TERMINATORS].word[t], D[D_TERMINATORS].len[t]);
        break;
      }
      default: break;
    }
    if (g->len == 0 || g->out[g->len - 1] != ' ') put(g, " ", 1);
  }
}

/* dists: "BEGIN name\nword|weight\n...\nEND\n" blocks (the format of dists.dss); `dists` is modified in place (cut into words).
 * out must hold size + 4096 bytes.  Returns 0, or a negative number for a malformed / missing distribution. */
int dbgen_text_pool(char* dists, char* out, int64_t size) {
  dist_t* D = (dist_t*)calloc(D_COUNT, sizeof(dist_t));
  int cur = -1;
  for (char* line = dists; line && *line;) {
    char* nl = strchr(line, '\n');
    if (nl) *nl = 0;
    if (!strncmp(line, "BEGIN ", 6)) {
      cur = -1;
      for (int k = 0; k < D_COUNT; k++)
        if (!strcmp(line + 6, NAMES[k])) cur = k;
    } else if (!strncmp(line, "END", 3)) {
      cur = -1;
    } else if (cur >= 0 && *line) {
      char* bar = strrchr(line, '|');
      if (!bar || D[cur].n >= MAX_ENTRIES) { free(D); return -1; }
      *bar = 0;
      dist_t* d = &D[cur];
      d->total += atoi(bar + 1);
      d->cum[d->n] = d->total;
      d->word[d->n] = line;
      d->len[d->n] = (int)strlen(line);
      d->n++;
    }
    line = nl ? nl + 1 : NULL;
  }
  for (int k = 0; k < D_COUNT; k++)
    if (k != D_ARTICLES && D[k].n == 0) { free(D); return -2 - k; }
  gen_t g = {933588178, out, 0};
  while (g.len < size) sentence(&g, D);
  free(D);
  return 0;
}
