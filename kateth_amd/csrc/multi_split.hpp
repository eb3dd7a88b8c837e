// How a group context (engine_multi.hip) cuts a batch over its members and merges their error records: plain host logic, no
// HIP -- kept apart so that the CPU test build (tests/hostmath) checks it against the Python restatement of the same rules
// (kateth_amd/dist.py shard_range / merge_first_error) without a GPU.
#pragma once
#include <stdint.h>

#include <vector>

namespace kzg {
namespace multi {

struct Share {
  uint32_t member;
  uint64_t first, count;
};

// Contiguous ranges of ceil(n / members) items in member order (blobs are independent: src/kzg/setup.rs:235-242).  A call with
// fewer items than members -- the single-item methods of the reference's API, made from many host threads at once -- hands out
// one item each starting at member `rotate % members`, so that concurrent small calls spread over the devices.
inline std::vector<Share> shares_of(uint64_t n, uint32_t members, uint32_t rotate) {
  std::vector<Share> out;
  if (n == 0 || members == 0) return out;
  if (n < members) {
    for (uint64_t i = 0; i < n; i++) out.push_back(Share{(uint32_t)((rotate + i) % members), i, 1});
    return out;
  }
  const uint64_t per = (n + members - 1) / members;
  for (uint32_t k = 0; k < members && (uint64_t)k * per < n; k++) {
    const uint64_t first = (uint64_t)k * per;
    out.push_back(Share{k, first, n - first < per ? n - first : per});
  }
  return out;
}

// First-error-wins order of the reference (src/kzg/setup.rs:259-271: every blob is parsed before any commitment, every
// commitment before any proof) from the shares' records err6 = {blob_idx, blob_code, commitment_idx, commitment_code, proof_idx,
// proof_code} with LOCAL indices (-1 = none): the lowest GLOBAL index of the first kind that has an error.  0 = no error.
inline int32_t merged_first_error(const std::vector<Share>& shares, const int32_t* err6) {
  for (int kind = 0; kind < 6; kind += 2) {
    int32_t code = 0;
    uint64_t best = ~(uint64_t)0;
    for (size_t j = 0; j < shares.size(); j++) {
      const int32_t local = err6[6 * j + kind];
      if (local < 0) continue;
      const uint64_t g = shares[j].first + (uint64_t)local;
      if (g < best) {
        best = g;
        code = err6[6 * j + kind + 1];
      }
    }
    if (code) return code;
  }
  return 0;
}

}  // namespace multi
}  // namespace kzg
