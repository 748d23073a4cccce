// interface shim (tests/faiss_shim/README.md): (list_no, offset) <-> 64-bit label
#pragma once
#include <cstdint>
namespace faiss {
inline uint64_t lo_build(uint64_t list_id, uint64_t offset) { return list_id << 32 | offset; }
inline uint64_t lo_listno(uint64_t lo) { return lo >> 32; }
inline uint64_t lo_offset(uint64_t lo) { return lo & 0xffffffff; }
}  // namespace faiss
