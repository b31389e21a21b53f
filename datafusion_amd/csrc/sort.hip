// sort.hip — placeholder, replaced by the real K11/K12 implementation
#include "internal.hpp"
using namespace dfgpu;
extern "C" {
int dfgpu_sort(dfgpu_table_t, const int*, const uint8_t*, const uint8_t*, int, int64_t, dfgpu_table_t*) { return guarded([] { throw Error("dfgpu_sort: not implemented"); }); }
}
