// aggregate.hip — placeholder, replaced by the real K6/K7 implementation
#include "internal.hpp"
using namespace dfgpu;
extern "C" {
int dfgpu_agg_create(int, const dfgpu_expr*, const char* const*, int, const dfgpu_agg_spec*, int, dfgpu_agg_t*) { return guarded([] { throw Error("dfgpu_agg_create: not implemented"); }); }
int dfgpu_agg_update(dfgpu_agg_t, dfgpu_table_t) { return guarded([] { throw Error("dfgpu_agg_update: not implemented"); }); }
int dfgpu_agg_emit(dfgpu_agg_t, dfgpu_table_t*) { return guarded([] { throw Error("dfgpu_agg_emit: not implemented"); }); }
int dfgpu_agg_free(dfgpu_agg_t) { return 0; }
}
