// grouped.hpp — rows moved into groups of their key (grouped.hip)
#pragma once
#include <vector>

#include "device.hpp"
#include "internal.hpp"

namespace dfgpu {

// group of a key: idx = key - offset (wrapping) must be < size; group = floor(idx * mul / 2^64) — `groups` equal stretches of the
// key range whatever its size (a shift would leave up to half of a power-of-two number of groups empty and the others twice as wide)
struct GroupSpec {
  uint64_t offset, size, mul;
};
inline GroupSpec group_spec(uint64_t offset, uint64_t size, int groups) {
  const unsigned __int128 m = (((unsigned __int128)(unsigned)groups) << 64) / (size ? size : 1);
  return GroupSpec{offset, size, m > (unsigned __int128)~0ull ? ~0ull : (uint64_t)m};
}
constexpr int GP_MAX_GROUPS = 2048;
constexpr int GP_MAX_COLS = 8;
struct GroupCols {
  const void* src[GP_MAX_COLS];
  void* dst[GP_MAX_COLS];
  int width[GP_MAX_COLS];
  int n;
  // the record form: plane c of a record = 32-bit word soff[c] of element i of src[c], elements sdw[c] words apart (an 8-byte column is two planes)
  int sdw[GP_MAX_COLS];
  int soff[GP_MAX_COLS];
};
struct GroupedRows {
  int P = 0;
  int64_t rows = 0;            // rows that took part
  BufPtr keys;                 // per grouped row (want_keys): the key widened to u64 — or, narrow_keys, key - offset as u32
  int key_width = 8;
  BufPtr dest;                 // u32 per INPUT row: its position in group order, ~0 = takes no part (want_dest)
  BufPtr bounds;               // u64 [P + 1] on the device: group g = positions bounds[g] .. bounds[g + 1]
  std::vector<BufPtr> cols;    // carried columns in group order
  BufPtr records;              // the record form (asked for and granted): u32 [rows][rec_dwords] = {key - offset, carried columns...}; keys / cols empty
  int rec_dwords = 0;
  std::vector<int> rec_off;    // per carried column: the 32-bit word of the record its value starts at (word 0 is the key)
};
// Rows of an integer key column moved into 2^nbits (<= GP_MAX_GROUPS) groups of their key's range (NULL keys, rows masked out by `row_mask` and keys
// outside [offset, offset + size) take no part).  Order inside a group is arbitrary.
GroupedRows group_rows_by_key(const KeyCol& key, int64_t n, const GroupSpec& gs, int nbits, const uint64_t* row_mask, bool want_keys, bool want_dest,
                              const std::vector<const void*>& carry_src, const std::vector<int>& carry_width, const char* what = nullptr, bool narrow_keys = false,
                              bool records = false);
// (a carried column of width 4 whose source is null carries the rows' numbers; `records`: 32-bit keys and carried 4- / 8-byte columns of at most 12 bytes together may
// leave as one record per row — GroupedRows::records — when the caller can read them that way)

}  // namespace dfgpu
