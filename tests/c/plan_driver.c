/*
 * plan_driver.c — the call sequences the DataFusion-side shim (shim/src/) performs, executed from PLAIN C through the C ABI of
 * include/dfgpu.h and nothing else (no Python, no C++, no torch): what tests/test_gpu_c_driver.py compiles with gcc, links against
 * libdfgpu.so and runs on the GPU box.  Test infrastructure, not product.
 *
 *   plan_driver join <case file>...
 *       GpuHashJoinExec::execute as shim/src/hash_join.rs documents it (HashJoinStream's states, hash_join/stream.rs:127-140):
 *         CollectBuildSide    dfgpu_join_builder_create; every build batch: dfgpu_table_import -> dfgpu_join_builder_push;
 *                             dfgpu_join_builder_finish                                            ONCE per join (OnceAsync, exec.rs:772)
 *         ProcessProbeBatch   per probe partition: dfgpu_table_import -> dfgpu_join_probe[_with_filter] -> dfgpu_table_export_batch
 *                             `batch_size` rows at a time
 *         ExhaustedProbeSide  after the LAST partition: dfgpu_join_emit_unmatched ONCE (exec.rs:1312-1330) -> export
 *       Prints the output rows; the Python side compares them with the reference's snapshot tests (tests/golden/hash_join_*.json).
 *
 *   plan_driver chain <scale factor>
 *       three adjacent GPU nodes — FilterExec -> AggregateExec -> SortExec — handing DEVICE tables to each other: the filter's
 *       output leaves as an ArrowDeviceArray (dfgpu_table_export_device) and enters the aggregate through
 *       dfgpu_table_import_device; dfgpu_metrics must report zero PCIe bytes for the whole chain.  A foreign producer is played
 *       too: a hand-made ArrowDeviceArray (release callback of this program) over device pointers, wrapped zero-copy.
 *
 *   plan_driver partial_final <scale factor> <partitions>
 *       the two-phase aggregate as shim/src/operators.rs runs it (aggregates/mod.rs:28-47): per input partition AggregateExec(Partial)
 *       [dfgpu_agg_create PARTIAL, one dfgpu_agg_update per batch, dfgpu_agg_emit = the state schema: keys, SUM -> sum, AVG -> count + sum,
 *       COUNT -> count, MIN -> value], RepartitionExec(Hash(keys, P)) of every partial output [dfgpu_partition], and per output
 *       partition AggregateExec(FinalPartitioned) fed partition p of every input [dfgpu_agg_create FINAL_PARTITIONED with the declared
 *       AVG type, one dfgpu_agg_update per input, dfgpu_agg_emit].  Then the same with GROUPING SETS ((flag), (status), (flag, status)):
 *       dfgpu_agg_create_grouping_sets PARTIAL -> dfgpu_partition on (keys, __grouping_id) -> a plain FINAL_PARTITIONED over them.
 *       Prints both results; the Python side compares them with the oracle's Single aggregates.
 *
 * Case file (text, whitespace separated):
 *   join_type N   null_equality N   null_aware N   table_mode N   build_batches K   probe_partitions P   batch_size B
 *   on NKEYS  l0 r0  l1 r1 ...
 *   build_out N c0 c1 ...        probe_out N c0 c1 ...
 *   filter NCOLS  (idx side)*NCOLS   OP LEFT RIGHT_IS_COL RIGHT     (NCOLS = 0: no JoinFilter; OP = dfgpu_expr_op of the comparison)
 *   table NCOLS NROWS  then per column: NAME v v v ... (N = NULL)      -- twice: left (build side), right (probe side)
 * Columns are Int32, as in the reference's snapshot tests (build_table_i32, hash_join/exec.rs:2880).
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dfgpu.h"

#define CHECK(call)                                                                                    \
  do {                                                                                                 \
    if ((call) != 0) {                                                                                 \
      fprintf(stderr, "%s:%d: %s failed: %s\n", __FILE__, __LINE__, #call, dfgpu_last_error());        \
      exit(2);                                                                                         \
    }                                                                                                  \
  } while (0)
#define REQUIRE(cond, msg)                                             \
  do {                                                                 \
    if (!(cond)) {                                                     \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, msg);         \
      exit(3);                                                         \
    }                                                                  \
  } while (0)

/* ------------------------------------------------------------------ host tables of Int32 columns */
typedef struct {
  int ncols;
  int64_t nrows;
  char** names;
  int32_t** values; /* [ncols][nrows] */
  uint8_t** valid;  /* [ncols][nrows]: 1 = not NULL */
} host_table;

static void read_token(FILE* f, char* buf, size_t n) {
  char fmt[16];
  snprintf(fmt, sizeof fmt, "%%%zus", n - 1);
  REQUIRE(fscanf(f, fmt, buf) == 1, "unexpected end of the case file");
}
static long read_long(FILE* f) {
  char b[64];
  read_token(f, b, sizeof b);
  return strtol(b, NULL, 10);
}
static void expect(FILE* f, const char* word) {
  char b[64];
  read_token(f, b, sizeof b);
  if (strcmp(b, word) != 0) {
    fprintf(stderr, "case file: expected '%s', found '%s'\n", word, b);
    exit(3);
  }
}
static host_table read_table(FILE* f) {
  host_table t;
  expect(f, "table");
  t.ncols = (int)read_long(f);
  t.nrows = read_long(f);
  t.names = (char**)calloc((size_t)t.ncols, sizeof(char*));
  t.values = (int32_t**)calloc((size_t)t.ncols, sizeof(int32_t*));
  t.valid = (uint8_t**)calloc((size_t)t.ncols, sizeof(uint8_t*));
  for (int c = 0; c < t.ncols; c++) {
    char b[128];
    read_token(f, b, sizeof b);
    t.names[c] = strdup(b);
    t.values[c] = (int32_t*)calloc((size_t)t.nrows + 1, 4);
    t.valid[c] = (uint8_t*)calloc((size_t)t.nrows + 1, 1);
    for (int64_t r = 0; r < t.nrows; r++) {
      read_token(f, b, sizeof b);
      if (strcmp(b, "N") == 0) continue;
      t.values[c][r] = (int32_t)strtol(b, NULL, 10);
      t.valid[c][r] = 1;
    }
  }
  return t;
}

/* ------------------------------------------------------------------ Arrow C Data structs made by hand (what arrow-rs `to_ffi` emits) */
typedef struct {
  void** owned; /* malloc'ed blocks to free */
  int n_owned;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowSchema** schema_children;
} priv;
static void release_array(struct ArrowArray* a) {
  priv* p = (priv*)a->private_data;
  for (int64_t i = 0; i < a->n_children; i++) {
    if (a->children[i]->release) a->children[i]->release(a->children[i]);
    free(a->children[i]);
  }
  for (int i = 0; i < p->n_owned; i++) free(p->owned[i]);
  free(p->owned);
  free((void*)p->buffers);
  free(p->children);
  free(p);
  a->release = NULL;
}
static void release_schema(struct ArrowSchema* s) {
  priv* p = (priv*)s->private_data;
  for (int64_t i = 0; i < s->n_children; i++) {
    if (s->children[i]->release) s->children[i]->release(s->children[i]);
    free(s->children[i]);
  }
  for (int i = 0; i < p->n_owned; i++) free(p->owned[i]);
  free(p->owned);
  free(p->schema_children);
  free(p);
  s->release = NULL;
}
/* rows [r0, r0 + n) of `t` as a struct array + schema (every column nullable Int32; validity only where a NULL is present) */
static void make_batch(const host_table* t, int64_t r0, int64_t n, struct ArrowArray* a, struct ArrowSchema* s) {
  memset(a, 0, sizeof *a);
  memset(s, 0, sizeof *s);
  priv* ap = (priv*)calloc(1, sizeof(priv));
  priv* sp = (priv*)calloc(1, sizeof(priv));
  ap->buffers = (const void**)calloc(1, sizeof(void*));
  ap->children = (struct ArrowArray**)calloc((size_t)t->ncols + 1, sizeof(void*));
  sp->schema_children = (struct ArrowSchema**)calloc((size_t)t->ncols + 1, sizeof(void*));
  for (int c = 0; c < t->ncols; c++) {
    struct ArrowArray* ca = (struct ArrowArray*)calloc(1, sizeof *ca);
    priv* cp = (priv*)calloc(1, sizeof(priv));
    cp->owned = (void**)calloc(2, sizeof(void*));
    cp->buffers = (const void**)calloc(2, sizeof(void*));
    int32_t* data = (int32_t*)calloc((size_t)n + 1, 4);
    memcpy(data, t->values[c] + r0, (size_t)n * 4);
    cp->owned[cp->n_owned++] = data;
    int64_t nulls = 0;
    for (int64_t r = 0; r < n; r++) nulls += t->valid[c][r0 + r] ? 0 : 1;
    uint8_t* bits = NULL;
    if (nulls) {
      bits = (uint8_t*)calloc((size_t)(n + 7) / 8 + 8, 1);
      for (int64_t r = 0; r < n; r++)
        if (t->valid[c][r0 + r]) bits[r >> 3] |= (uint8_t)(1u << (r & 7));
      cp->owned[cp->n_owned++] = bits;
    }
    cp->buffers[0] = bits;
    cp->buffers[1] = data;
    ca->length = n;
    ca->null_count = nulls;
    ca->n_buffers = 2;
    ca->buffers = cp->buffers;
    ca->release = release_array;
    ca->private_data = cp;
    ap->children[c] = ca;
    struct ArrowSchema* cs = (struct ArrowSchema*)calloc(1, sizeof *cs);
    priv* csp = (priv*)calloc(1, sizeof(priv));
    cs->format = "i";
    cs->name = t->names[c];
    cs->flags = 2; /* ARROW_FLAG_NULLABLE */
    cs->release = release_schema;
    cs->private_data = csp;
    sp->schema_children[c] = cs;
  }
  a->length = n;
  a->n_buffers = 1;
  a->buffers = ap->buffers;
  a->n_children = t->ncols;
  a->children = ap->children;
  a->release = release_array;
  a->private_data = ap;
  s->format = "+s";
  s->name = "";
  s->n_children = t->ncols;
  s->children = sp->schema_children;
  s->release = release_schema;
  s->private_data = sp;
}
static dfgpu_table_t import_rows(const host_table* t, int64_t r0, int64_t n) {
  struct ArrowArray a;
  struct ArrowSchema s;
  make_batch(t, r0, n, &a, &s);
  dfgpu_table_t out = NULL;
  CHECK(dfgpu_table_import(&a, &s, &out)); /* consumes both structs */
  REQUIRE(a.release == NULL && s.release == NULL, "dfgpu_table_import did not release its arguments");
  return out;
}

/* ------------------------------------------------------------------ printing exported batches */
static int bit(const void* bits, int64_t i) { return (((const uint8_t*)bits)[i >> 3] >> (i & 7)) & 1; }
static void print_value(const struct ArrowArray* c, const char* fmt, int64_t row) {
  const int64_t i = c->offset + row;
  if (c->buffers[0] && !bit(c->buffers[0], i)) {
    printf("NULL");
    return;
  }
  if (strcmp(fmt, "i") == 0 || strcmp(fmt, "tdD") == 0) printf("%" PRId32, ((const int32_t*)c->buffers[1])[i]);
  else if (strcmp(fmt, "l") == 0) printf("%" PRId64, ((const int64_t*)c->buffers[1])[i]);
  else if (strcmp(fmt, "C") == 0) printf("%u", (unsigned)((const uint8_t*)c->buffers[1])[i]);
  else if (strcmp(fmt, "b") == 0) printf("%s", bit(c->buffers[1], i) ? "true" : "false");
  else if (strcmp(fmt, "g") == 0) printf("%.17g", ((const double*)c->buffers[1])[i]);
  else if (strncmp(fmt, "d:", 2) == 0) {
    /* Decimal128 as its unscaled value: the tests' sums stay inside 64 bits, the high word must be the sign extension */
    const int64_t* w = (const int64_t*)c->buffers[1] + 2 * i;
    REQUIRE(w[1] == (w[0] < 0 ? -1 : 0), "decimal value beyond 64 bits");
    printf("%" PRId64, w[0]);
  } else {
    fprintf(stderr, "print_value: unexpected format '%s'\n", fmt);
    exit(3);
  }
}
/* export `t` batch_size rows at a time, as a GPU node's poll_next does, and print every row */
static int64_t export_and_print(dfgpu_table_t t, int64_t batch_size, int print_header) {
  int64_t n = 0;
  CHECK(dfgpu_table_num_rows(t, &n));
  int ncols = 0;
  CHECK(dfgpu_table_num_columns(t, &ncols));
  if (print_header) {
    printf("columns");
    for (int c = 0; c < ncols; c++) {
      dfgpu_column_view v;
      CHECK(dfgpu_table_column(t, c, &v));
      printf(" %s", v.name);
    }
    printf("\n");
  }
  for (int64_t off = 0; off < n; off += batch_size) {
    const int64_t len = n - off < batch_size ? n - off : batch_size;
    struct ArrowArray a;
    struct ArrowSchema s;
    CHECK(dfgpu_table_export_batch(t, off, len, &a, &s));
    REQUIRE(a.length == len && a.n_children == ncols && s.n_children == ncols, "exported batch has the wrong shape");
    for (int64_t r = 0; r < len; r++) {
      printf("row");
      for (int c = 0; c < ncols; c++) {
        printf(" ");
        print_value(a.children[c], s.children[c]->format, r);
      }
      printf("\n");
    }
    a.release(&a);
    s.release(&s);
  }
  return n;
}

/* ------------------------------------------------------------------ mode: join */
static int builds_side_rows(int join_type) { /* the join types whose build rows are reported after the probe side is exhausted */
  return join_type == DFGPU_JOIN_LEFT || join_type == DFGPU_JOIN_FULL || join_type == DFGPU_JOIN_LEFT_SEMI || join_type == DFGPU_JOIN_LEFT_ANTI ||
         join_type == DFGPU_JOIN_LEFT_MARK;
}
static void run_join_case(const char* path) {
  FILE* f = fopen(path, "r");
  REQUIRE(f != NULL, "cannot open the case file");
  expect(f, "join_type");        const int join_type = (int)read_long(f);
  expect(f, "null_equality");    const int null_equality = (int)read_long(f);
  expect(f, "null_aware");       const int null_aware = (int)read_long(f);
  expect(f, "table_mode");       const int table_mode = (int)read_long(f);
  expect(f, "build_batches");    const int build_batches = (int)read_long(f);
  expect(f, "probe_partitions"); const int probe_partitions = (int)read_long(f);
  expect(f, "batch_size");       const int64_t batch_size = read_long(f);
  expect(f, "on");
  const int nkeys = (int)read_long(f);
  int lk[8], rk[8];
  REQUIRE(nkeys >= 1 && nkeys <= 8, "bad key count");
  for (int i = 0; i < nkeys; i++) { lk[i] = (int)read_long(f); rk[i] = (int)read_long(f); }
  expect(f, "build_out");
  const int n_bo = (int)read_long(f);
  int bo[64];
  for (int i = 0; i < n_bo; i++) bo[i] = (int)read_long(f);
  expect(f, "probe_out");
  const int n_po = (int)read_long(f);
  int po[64];
  for (int i = 0; i < n_po; i++) po[i] = (int)read_long(f);
  expect(f, "filter");
  const int n_fcols = (int)read_long(f);
  int32_t fidx[16], fside[16];
  int f_op = 0, f_left = 0, f_right_is_col = 0;
  long f_right = 0;
  for (int i = 0; i < n_fcols; i++) { fidx[i] = (int32_t)read_long(f); fside[i] = (int32_t)read_long(f); }
  if (n_fcols) { f_op = (int)read_long(f); f_left = (int)read_long(f); f_right_is_col = (int)read_long(f); f_right = read_long(f); }
  host_table left = read_table(f), right = read_table(f);
  fclose(f);

  /* ---- CollectBuildSide: the build child's batches stream into the builder; ONE table per join */
  dfgpu_join_options opts;
  opts.perfect_hash_join_small_build_threshold = 1024;
  opts.perfect_hash_join_min_key_density = DFGPU_DEFAULT_MIN_KEY_DENSITY;
  opts.table_mode = table_mode;
  opts.force_hash_collisions = 0;
  opts.probe_mode = 0; /* the reference's order: an ancestor may observe it */
  opts.null_aware = null_aware;
  dfgpu_join_builder_t builder = NULL;
  CHECK(dfgpu_join_builder_create(lk, nkeys, null_equality, &opts, &builder));
  for (int b = 0; b < build_batches; b++) {
    const int64_t r0 = left.nrows * b / build_batches, r1 = left.nrows * (b + 1) / build_batches;
    dfgpu_table_t batch = import_rows(&left, r0, r1 - r0);
    CHECK(dfgpu_join_builder_push(builder, batch));
    CHECK(dfgpu_table_free(batch)); /* the builder holds its own reference */
  }
  dfgpu_join_t ht = NULL;
  CHECK(dfgpu_join_builder_finish(builder, &ht)); /* consumes the builder */

  /* ---- the JoinFilter, lowered as shim/src/expr.rs lowers a BinaryExpr over the intermediate batch's columns */
  dfgpu_expr_node nodes[3];
  dfgpu_join_filter jf;
  memset(nodes, 0, sizeof nodes);
  memset(&jf, 0, sizeof jf);
  if (n_fcols) {
    nodes[0].op = DFGPU_EXPR_COLUMN; nodes[0].column = f_left; nodes[0].left = nodes[0].right = -1;
    if (f_right_is_col) {
      nodes[1].op = DFGPU_EXPR_COLUMN; nodes[1].column = (int32_t)f_right; nodes[1].left = nodes[1].right = -1;
    } else {
      nodes[1].op = DFGPU_EXPR_LITERAL; nodes[1].column = -1; nodes[1].left = nodes[1].right = -1;
      nodes[1].field.type = DFGPU_INT32; nodes[1].field.nullable = 1;
      nodes[1].lit_lo = (uint64_t)(int64_t)f_right; nodes[1].lit_hi = f_right < 0 ? ~(uint64_t)0 : 0;
    }
    nodes[2].op = f_op; nodes[2].column = -1; nodes[2].left = 0; nodes[2].right = 1;
    jf.expression.nodes = nodes; jf.expression.n_nodes = 3; jf.expression.root = 2; jf.expression.string_pool = NULL;
    jf.column_index = fidx; jf.column_side = fside; jf.n_columns = n_fcols;
  }

  /* ---- one probe call per probe partition against the shared table */
  int header = 1;
  int64_t total = 0;
  for (int p = 0; p < probe_partitions; p++) {
    const int64_t r0 = right.nrows * p / probe_partitions, r1 = right.nrows * (p + 1) / probe_partitions;
    dfgpu_table_t probe = import_rows(&right, r0, r1 - r0);
    dfgpu_table_t out = NULL;
    if (n_fcols) CHECK(dfgpu_join_probe_with_filter(ht, probe, rk, join_type, &jf, bo, n_bo, po, n_po, &out));
    else CHECK(dfgpu_join_probe(ht, probe, rk, join_type, bo, n_bo, po, n_po, &out));
    if (!builds_side_rows(join_type) || join_type == DFGPU_JOIN_LEFT || join_type == DFGPU_JOIN_FULL) {
      total += export_and_print(out, batch_size, header);   /* matched pairs / probe-side rows of this partition */
      header = 0;
    }
    CHECK(dfgpu_table_free(out));
    CHECK(dfgpu_table_free(probe));
  }
  /* ---- ExhaustedProbeSide: the last partition reports the build rows by their visited marks, once */
  if (builds_side_rows(join_type)) {
    dfgpu_field pf[64];
    const char* pn[64];
    int n_tail_probe = 0;
    if (join_type == DFGPU_JOIN_LEFT || join_type == DFGPU_JOIN_FULL) {
      for (int i = 0; i < n_po; i++) {
        pf[i].type = DFGPU_INT32; pf[i].precision = 0; pf[i].scale = 0; pf[i].nullable = 1;
        pn[i] = right.names[po[i]];
      }
      n_tail_probe = n_po;
    }
    dfgpu_table_t tail = NULL;
    CHECK(dfgpu_join_emit_unmatched(ht, join_type, bo, n_bo, pf, pn, n_tail_probe, &tail));
    total += export_and_print(tail, batch_size, header);
    CHECK(dfgpu_table_free(tail));
  }
  dfgpu_join_info info;
  CHECK(dfgpu_join_get_info(ht, &info));
  printf("info build_rows %" PRId64 " probe_rows %" PRId64 " table_kind %d rows_printed %" PRId64 "\n", info.build_rows, info.probe_rows, info.table_kind, total);
  REQUIRE(info.build_rows == left.nrows && info.probe_rows == right.nrows, "join info does not add up");
  CHECK(dfgpu_join_free(ht));
}

/* ------------------------------------------------------------------ mode: chain */
static void foreign_release(struct ArrowArray* a) { /* the release callback of a producer that is not this library */
  if (a->private_data) (*(int*)a->private_data)++;
  a->release = NULL;
}
static dfgpu_expr_node col_node(int column) {
  dfgpu_expr_node n;
  memset(&n, 0, sizeof n);
  n.op = DFGPU_EXPR_COLUMN; n.column = column; n.left = n.right = -1;
  return n;
}
static int find_column(dfgpu_table_t t, const char* name) {
  int ncols = 0;
  CHECK(dfgpu_table_num_columns(t, &ncols));
  for (int c = 0; c < ncols; c++) {
    dfgpu_column_view v;
    CHECK(dfgpu_table_column(t, c, &v));
    if (strcmp(v.name, name) == 0) return c;
  }
  fprintf(stderr, "no column %s\n", name);
  exit(3);
}
static void run_chain(double sf) {
  dfgpu_table_t lineitem = NULL;
  CHECK(dfgpu_tpch_lineitem(sf, 0, -1, 0, &lineitem)); /* a device-resident scan (the chunk cache's role in a real plan) */
  CHECK(dfgpu_metrics_reset());

  /* node 1: FilterExec l_shipdate <= 1998-09-02 (day 10471) */
  const int c_ship = find_column(lineitem, "l_shipdate");
  dfgpu_expr_node pn[3];
  pn[0] = col_node(c_ship);
  memset(&pn[1], 0, sizeof pn[1]);
  pn[1].op = DFGPU_EXPR_LITERAL; pn[1].column = -1; pn[1].left = pn[1].right = -1; pn[1].field.type = DFGPU_DATE32; pn[1].field.nullable = 1; pn[1].lit_lo = 10471;
  memset(&pn[2], 0, sizeof pn[2]);
  pn[2].op = DFGPU_EXPR_LE; pn[2].column = -1; pn[2].left = 0; pn[2].right = 1;
  dfgpu_expr pred; pred.nodes = pn; pred.n_nodes = 3; pred.root = 2; pred.string_pool = NULL;
  dfgpu_table_t filtered = NULL;
  CHECK(dfgpu_filter(lineitem, &pred, NULL, 0, &filtered));

  /* the hand-off: node 1's stream yields an ArrowDeviceArray, node 2 takes it — no host copy */
  struct ArrowDeviceArray dev;
  struct ArrowSchema dev_schema;
  CHECK(dfgpu_table_export_device(filtered, &dev, &dev_schema));
  REQUIRE(dev.device_type == ARROW_DEVICE_ROCM && dev.sync_event == NULL, "device array is not tagged ROCm / ready");
  int64_t n_filtered = 0;
  CHECK(dfgpu_table_num_rows(filtered, &n_filtered));
  REQUIRE(dev.array.length == n_filtered, "device array length");
  dfgpu_column_view v0;
  CHECK(dfgpu_table_column(filtered, 0, &v0));
  REQUIRE(dev.array.children[0]->buffers[1] == v0.data, "the device array does not point into the table's own HBM buffer");
  CHECK(dfgpu_table_free(filtered)); /* the array keeps the buffers alive */
  dfgpu_table_t agg_in = NULL;
  CHECK(dfgpu_table_import_device(&dev, &dev_schema, &agg_in));
  REQUIRE(dev.array.release == NULL && dev_schema.release == NULL, "dfgpu_table_import_device did not consume its arguments");
  dfgpu_column_view v1;
  CHECK(dfgpu_table_column(agg_in, 0, &v1));
  REQUIRE(v1.data == v0.data, "the imported table does not share the exported buffers");

  /* a FOREIGN producer played by hand: an ArrowDeviceArray over two of the table's device columns, with a release callback of
   * this program's own — the library must wrap it zero-copy and call release when the wrapping table is freed */
  {
    static int released = 0;
    const int c_key = find_column(agg_in, "l_orderkey"), c_flag = find_column(agg_in, "l_returnflag");
    dfgpu_column_view vk, vf;
    CHECK(dfgpu_table_column(agg_in, c_key, &vk));
    CHECK(dfgpu_table_column(agg_in, c_flag, &vf));
    struct ArrowArray* kids[2];
    struct ArrowSchema* skids[2];
    static const void* kb[2][2];
    static struct ArrowArray ka[2];
    static struct ArrowSchema ks[2];
    static struct ArrowSchema root_schema;
    memset(ka, 0, sizeof ka);
    memset(ks, 0, sizeof ks);
    kb[0][0] = NULL; kb[0][1] = vk.data;
    kb[1][0] = NULL; kb[1][1] = vf.data;
    for (int i = 0; i < 2; i++) {
      ka[i].length = n_filtered; ka[i].n_buffers = 2; ka[i].buffers = kb[i]; ka[i].release = foreign_release; ka[i].private_data = NULL;
      ks[i].format = i == 0 ? "l" : "C"; ks[i].name = i == 0 ? "k" : "flag"; ks[i].flags = 2; ks[i].release = NULL;
      kids[i] = &ka[i]; skids[i] = &ks[i];
    }
    struct ArrowDeviceArray fa;
    memset(&fa, 0, sizeof fa);
    static const void* root_buffers[1] = {NULL};
    fa.array.length = n_filtered; fa.array.n_buffers = 1; fa.array.buffers = root_buffers; fa.array.n_children = 2; fa.array.children = kids;
    fa.array.release = foreign_release; fa.array.private_data = &released;
    fa.device_id = dev.device_id; fa.device_type = ARROW_DEVICE_ROCM; fa.sync_event = NULL;
    memset(&root_schema, 0, sizeof root_schema);
    root_schema.format = "+s"; root_schema.name = ""; root_schema.n_children = 2; root_schema.children = skids; root_schema.release = NULL;
    dfgpu_table_t wrapped = NULL;
    CHECK(dfgpu_table_import_device(&fa, &root_schema, &wrapped));
    dfgpu_column_view wk;
    CHECK(dfgpu_table_column(wrapped, 0, &wk));
    REQUIRE(wk.data == vk.data && wk.length == n_filtered && wk.field.type == DFGPU_INT64, "foreign device array was not wrapped zero-copy");
    /* usable by an operator: COUNT(*) GROUP BY flag over the wrapped table */
    dfgpu_expr_node g = col_node(1);
    dfgpu_expr ge; ge.nodes = &g; ge.n_nodes = 1; ge.root = 0; ge.string_pool = NULL;
    const char* gname = "flag";
    dfgpu_agg_spec cnt; memset(&cnt, 0, sizeof cnt);
    cnt.func = DFGPU_AGG_COUNT; cnt.has_arg = 0; cnt.name = "n";
    dfgpu_agg_t h = NULL;
    CHECK(dfgpu_agg_create(DFGPU_AGG_SINGLE, &ge, &gname, 1, &cnt, 1, &h));
    CHECK(dfgpu_agg_update(h, wrapped));
    dfgpu_table_t counts = NULL;
    CHECK(dfgpu_agg_emit(h, &counts));
    CHECK(dfgpu_agg_free(h));
    int64_t ngroups = 0;
    CHECK(dfgpu_table_num_rows(counts, &ngroups));
    REQUIRE(ngroups >= 1 && ngroups <= 3, "aggregate over the wrapped foreign array");
    CHECK(dfgpu_table_free(counts));
    REQUIRE(released == 0, "the foreign array was released while a table still points into it");
    CHECK(dfgpu_table_free(wrapped));
    REQUIRE(released == 1, "the foreign array's release callback did not run with the last reference");
  }

  /* node 2: AggregateExec(Single) GROUP BY l_returnflag, l_linestatus: SUM(l_quantity), SUM(l_extendedprice), COUNT(*) */
  dfgpu_expr_node gn[2];
  gn[0] = col_node(find_column(agg_in, "l_returnflag"));
  gn[1] = col_node(find_column(agg_in, "l_linestatus"));
  dfgpu_expr group_by[2];
  for (int i = 0; i < 2; i++) { group_by[i].nodes = &gn[i]; group_by[i].n_nodes = 1; group_by[i].root = 0; group_by[i].string_pool = NULL; }
  const char* group_names[2] = {"l_returnflag", "l_linestatus"};
  dfgpu_expr_node an[2];
  an[0] = col_node(find_column(agg_in, "l_quantity"));
  an[1] = col_node(find_column(agg_in, "l_extendedprice"));
  dfgpu_agg_spec aggs[3];
  memset(aggs, 0, sizeof aggs);
  for (int i = 0; i < 2; i++) {
    aggs[i].func = DFGPU_AGG_SUM; aggs[i].has_arg = 1;
    aggs[i].arg.nodes = &an[i]; aggs[i].arg.n_nodes = 1; aggs[i].arg.root = 0; aggs[i].arg.string_pool = NULL;
  }
  aggs[0].name = "sum_qty"; aggs[1].name = "sum_base_price";
  aggs[2].func = DFGPU_AGG_COUNT; aggs[2].has_arg = 0; aggs[2].name = "count_order";
  dfgpu_agg_t agg = NULL;
  CHECK(dfgpu_agg_create(DFGPU_AGG_SINGLE, group_by, group_names, 2, aggs, 3, &agg));
  CHECK(dfgpu_agg_update(agg, agg_in));
  dfgpu_table_t grouped = NULL;
  CHECK(dfgpu_agg_emit(agg, &grouped));
  CHECK(dfgpu_agg_free(agg));
  CHECK(dfgpu_table_free(agg_in));

  /* node 3: SortExec by (l_returnflag, l_linestatus) */
  const int keys[2] = {0, 1};
  const uint8_t desc[2] = {0, 0}, nulls_first[2] = {0, 0};
  dfgpu_table_t sorted = NULL;
  CHECK(dfgpu_sort(grouped, keys, desc, nulls_first, 2, -1, &sorted));
  CHECK(dfgpu_table_free(grouped));

  dfgpu_metrics m;
  CHECK(dfgpu_metrics_get(&m));
  printf("metrics calls %" PRId64 " h2d_bytes %" PRId64 " d2h_bytes %" PRId64 " rows_filtered %" PRId64 "\n", m.calls, m.h2d_bytes, m.d2h_bytes, n_filtered);
  REQUIRE(m.h2d_bytes == 0 && m.d2h_bytes == 0, "the three-node chain moved table bytes across PCIe");
  export_and_print(sorted, 8192, 1); /* only the final result crosses to the host */
  CHECK(dfgpu_table_free(sorted));
  CHECK(dfgpu_table_free(lineitem));
}
/* ------------------------------------------------------------------ mode: partial_final */
static dfgpu_expr expr_of(dfgpu_expr_node* n) {
  dfgpu_expr e;
  e.nodes = n; e.n_nodes = 1; e.root = 0; e.string_pool = NULL;
  return e;
}
/* Partial aggregates of `n_in` input partitions -> hash repartition on the first n_key output columns -> FinalPartitioned per output
 * partition; prints the rows of all output partitions under `title` */
static void two_phase(const char* title, dfgpu_agg_t* partials, int n_in, int n_key, int nparts, const dfgpu_expr* group_by, const char* const* key_names,
                      const dfgpu_agg_spec* final_aggs, int n_aggs) {
  dfgpu_table_t* parts = (dfgpu_table_t*)calloc((size_t)n_in * nparts, sizeof(dfgpu_table_t));
  int key_cols[8];
  for (int k = 0; k < n_key; k++) key_cols[k] = k;
  for (int i = 0; i < n_in; i++) {
    dfgpu_table_t state = NULL;
    CHECK(dfgpu_agg_emit(partials[i], &state));          /* AggregateExec(Partial)'s output: keys + state columns */
    CHECK(dfgpu_agg_free(partials[i]));
    CHECK(dfgpu_partition(state, key_cols, n_key, nparts, parts + (size_t)i * nparts));   /* RepartitionExec(Hash(keys, nparts)) */
    CHECK(dfgpu_table_free(state));
  }
  printf("%s\n", title);
  int64_t printed = 0;
  for (int p = 0; p < nparts; p++) {
    dfgpu_agg_t fin = NULL;
    CHECK(dfgpu_agg_create(DFGPU_AGG_FINAL_PARTITIONED, group_by, key_names, n_key, final_aggs, n_aggs, &fin));
    for (int i = 0; i < n_in; i++) {                      /* partition p of every input partition arrives at output partition p */
      CHECK(dfgpu_agg_update(fin, parts[(size_t)i * nparts + p]));
      CHECK(dfgpu_table_free(parts[(size_t)i * nparts + p]));
    }
    dfgpu_table_t out = NULL;
    CHECK(dfgpu_agg_emit(fin, &out));
    CHECK(dfgpu_agg_free(fin));
    printed += export_and_print(out, 8192, p == 0);
    CHECK(dfgpu_table_free(out));
  }
  printf("rows %" PRId64 "\n", printed);
  free(parts);
}
static void run_partial_final(double sf, int nparts) {
  REQUIRE(nparts >= 1 && nparts <= 8, "1..8 partitions");
  dfgpu_table_t lineitem = NULL;
  CHECK(dfgpu_tpch_lineitem(sf, 0, -1, 0, &lineitem));
  int64_t n = 0;
  CHECK(dfgpu_table_num_rows(lineitem, &n));
  const int n_in = 3;                                     /* three input partitions, each fed in two batches */
  dfgpu_expr_node gn[2], an[2], nulls[2];
  gn[0] = col_node(find_column(lineitem, "l_returnflag"));
  gn[1] = col_node(find_column(lineitem, "l_linestatus"));
  an[0] = col_node(find_column(lineitem, "l_quantity"));
  an[1] = col_node(find_column(lineitem, "l_extendedprice"));
  dfgpu_expr group_by[2] = {expr_of(&gn[0]), expr_of(&gn[1])};
  const char* key_names[3] = {"l_returnflag", "l_linestatus", "__grouping_id"};
  dfgpu_agg_spec aggs[4];
  memset(aggs, 0, sizeof aggs);
  aggs[0].func = DFGPU_AGG_SUM; aggs[0].has_arg = 1; aggs[0].arg = expr_of(&an[0]); aggs[0].name = "sum_qty";
  aggs[1].func = DFGPU_AGG_AVG; aggs[1].has_arg = 1; aggs[1].arg = expr_of(&an[1]); aggs[1].name = "avg_price";
  aggs[2].func = DFGPU_AGG_COUNT; aggs[2].has_arg = 0; aggs[2].name = "count_order";
  aggs[3].func = DFGPU_AGG_MIN; aggs[3].has_arg = 1; aggs[3].arg = expr_of(&an[0]); aggs[3].name = "min_qty";
  /* what AggregateFunctionExpr::return_field carries to the Final node: AVG(Decimal128(15,2)) -> Decimal128(19,6) (average.rs:219-252) */
  dfgpu_agg_spec final_aggs[4];
  memcpy(final_aggs, aggs, sizeof aggs);
  final_aggs[1].return_field.type = DFGPU_DECIMAL128; final_aggs[1].return_field.precision = 19; final_aggs[1].return_field.scale = 6; final_aggs[1].return_field.nullable = 1;

  dfgpu_agg_t partials[3];
  for (int pass = 0; pass < 2; pass++) {
    for (int i = 0; i < n_in; i++) {
      if (pass == 0) {
        CHECK(dfgpu_agg_create(DFGPU_AGG_PARTIAL, group_by, key_names, 2, aggs, 4, &partials[i]));
      } else {
        memset(nulls, 0, sizeof nulls);
        for (int g = 0; g < 2; g++) {                      /* the typed NULL literal of each key (PhysicalGroupBy::null_expr) */
          nulls[g].op = DFGPU_EXPR_LITERAL; nulls[g].column = -1; nulls[g].left = nulls[g].right = -1;
          nulls[g].field.type = DFGPU_UINT8; nulls[g].field.nullable = 1; nulls[g].is_null = 1;
        }
        dfgpu_expr null_by[2] = {expr_of(&nulls[0]), expr_of(&nulls[1])};
        const uint8_t sets[3 * 2] = {0, 1, /* (flag) */ 1, 0, /* (status) */ 0, 0 /* (flag, status) */};
        CHECK(dfgpu_agg_create_grouping_sets(DFGPU_AGG_PARTIAL, group_by, null_by, key_names, 2, sets, 3, aggs, 4, &partials[i]));
      }
      const int64_t lo = n * i / n_in, hi = n * (i + 1) / n_in, mid = lo + (hi - lo) / 2;
      const int64_t cut[3] = {lo, mid, hi};
      for (int b = 0; b < 2; b++) {                        /* the partition's input stream: two batches */
        dfgpu_table_t batch = NULL;
        CHECK(dfgpu_table_slice(lineitem, cut[b], cut[b + 1] - cut[b], &batch));
        CHECK(dfgpu_agg_update(partials[i], batch));
        CHECK(dfgpu_table_free(batch));
      }
    }
    /* a Final node's group expressions are the leading columns of the partial state: Column(0), Column(1) [, Column(2) = __grouping_id] */
    dfgpu_expr_node fk[3] = {col_node(0), col_node(1), col_node(2)};
    dfgpu_expr final_keys[3] = {expr_of(&fk[0]), expr_of(&fk[1]), expr_of(&fk[2])};
    if (pass == 0) two_phase("two_phase", partials, n_in, 2, nparts, final_keys, key_names, final_aggs, 4);
    else two_phase("grouping_sets", partials, n_in, 3, nparts, final_keys, key_names, final_aggs, 4);   /* the Final groups by keys + __grouping_id */
  }
  CHECK(dfgpu_table_free(lineitem));
}
int main(int argc, char** argv) {
  REQUIRE(argc >= 3, "usage: plan_driver join <case>... | plan_driver chain <sf> | plan_driver partial_final <sf> <partitions>");
  REQUIRE(dfgpu_abi_version() == DFGPU_ABI_VERSION, "libdfgpu.so was built from another header");
  const int device = 0;
  CHECK(dfgpu_init(&device, 1));
  if (strcmp(argv[1], "join") == 0) {
    for (int i = 2; i < argc; i++) {
      printf("case %s\n", argv[i]);
      run_join_case(argv[i]);
      printf("end\n");
    }
  } else if (strcmp(argv[1], "chain") == 0) {
    run_chain(strtod(argv[2], NULL));
  } else if (strcmp(argv[1], "partial_final") == 0) {
    REQUIRE(argc >= 4, "usage: plan_driver partial_final <sf> <partitions>");
    run_partial_final(strtod(argv[2], NULL), atoi(argv[3]));
  } else {
    REQUIRE(0, "unknown mode");
  }
  CHECK(dfgpu_shutdown());
  return 0;
}
