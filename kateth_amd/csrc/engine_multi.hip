// Multi-GPU behind the C ABI (`kzg_ctx_create(g1, g2, devices[], ndev, &ctx)`).
//
// A GROUP context is member 0 -- a complete single-device context -- plus one peer context per further listed device, each
// holding the full tables (the setup is 4096 + 65 points: replicated, never sharded).  Blobs are independent in all three
// operations (the reference walks them one by one: src/kzg/setup.rs:235-242), so the host-buffer entry points cut a batch
// into contiguous ranges, one per member, run the single-device implementation of each range on a host thread of its own and
// let it write straight into the caller's buffers: no collective, no gather.  The only cross-member step is the end of batch
// verification (src/kzg/setup.rs:152-160): every member returns its transcript root and first-error records, the roots of
// ALL members seed the batch challenge, every member returns its two partial sums with GLOBAL powers r^i, and one member sums
// the partials and runs the single pairing check -- the same phase1 / roots / phase2 / finish protocol that
// kateth_amd/dist.py drives across processes with RCCL, here inside one process.
#include "engine_internal.hpp"
#include "multi_split.hpp"

using kzg::multi::Share;
using kzg::multi::merged_first_error;

namespace {

inline const kzg_ctx* member_of(const kzg_ctx* ctx, uint32_t k) { return k == 0 ? ctx : ctx->peers[k - 1]; }

// the shares of a call over this group's members (multi_split.hpp); calls smaller than the group start at a rotating member
std::vector<Share> shares_of(const kzg_ctx* ctx, uint64_t n) {
  const uint32_t S = 1u + (uint32_t)ctx->peers.size();
  const uint32_t rotate = (n && n < S) ? ctx->rr.fetch_add((uint32_t)n, std::memory_order_relaxed) : 0u;
  return kzg::multi::shares_of(n, S, rotate);
}

}  // namespace

int32_t group_create(const uint8_t* g1_lagrange, const uint8_t* g2_monomial, const kzg_config* cfg, kzg_ctx** out) {
  *out = nullptr;
  const int32_t visible = kzg_device_count();
  if (visible < 0) return visible;
  std::vector<int> devices;
  if (cfg->ndev == KZG_ALL_DEVICES) {
    for (int d = 0; d < visible; d++) devices.push_back(d);
  } else {
    if (!cfg->devices) return fail(KZG_FAIL_ARGUMENT, "kzg_config.devices is null with ndev != KZG_ALL_DEVICES");
    if (cfg->ndev > 64) return fail(KZG_FAIL_ARGUMENT, "kzg_config.ndev: at most 64 members");
    for (uint32_t k = 0; k < cfg->ndev; k++) {
      if (cfg->devices[k] < 0 || cfg->devices[k] >= visible) return fail(KZG_FAIL_ARGUMENT, "device ordinal out of range in kzg_config.devices");
      devices.push_back(cfg->devices[k]);
    }
  }
  // every member decodes the setup and builds its tables on its own device, all at once (one host thread per member)
  std::vector<kzg_ctx*> members(devices.size(), nullptr);
  const int32_t rc = run_on_helpers((uint32_t)devices.size(), [&](uint32_t k) -> int32_t {
    return ctx_create_single(g1_lagrange, g2_monomial, cfg, devices[k], &members[k]);
  });
  if (rc) {
    const ErrorSnapshot keep = error_snapshot();
    for (kzg_ctx* m : members)
      if (m) kzg_ctx_destroy(m);
    error_publish(keep);
    return rc;
  }
  kzg_ctx* head = members[0];
  head->peers.assign(members.begin() + 1, members.end());
  *out = head;
  return 0;
}

int32_t multi_commit(const kzg_ctx* ctx, const uint8_t* blobs, uint64_t n, uint8_t* out48, uint8_t* out_affine96, int32_t* status) {
  if (!ctx || (n && (!blobs || (!out48 && !out_affine96) || !status))) return fail(KZG_FAIL_ARGUMENT, "null argument");
  const std::vector<Share> shares = shares_of(ctx, n);
  return run_on_helpers((uint32_t)shares.size(), [&](uint32_t j) -> int32_t {
    const Share& sh = shares[j];
    return commit_host(member_of(ctx, sh.member), blobs + sh.first * (size_t)KZG_BYTES_PER_BLOB, sh.count, out48 ? out48 + sh.first * 48 : nullptr,
                       out_affine96 ? out_affine96 + sh.first * 96 : nullptr, status + sh.first);
  });
}

int32_t multi_proof(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* side, size_t side_bytes, bool side_is_commitment, uint64_t n, uint8_t* out48,
                    uint8_t* out_affine96, uint8_t* out_y32, int32_t* status) {
  const std::vector<Share> shares = shares_of(ctx, n);
  return run_on_helpers((uint32_t)shares.size(), [&](uint32_t j) -> int32_t {
    const Share& sh = shares[j];
    return proof_host(member_of(ctx, sh.member), blobs + sh.first * (size_t)KZG_BYTES_PER_BLOB, side + sh.first * side_bytes, side_bytes, side_is_commitment,
                      sh.count, out48 ? out48 + sh.first * 48 : nullptr, out_affine96 ? out_affine96 + sh.first * 96 : nullptr,
                      out_y32 ? out_y32 + sh.first * 32 : nullptr, status + sh.first);
  });
}

int32_t multi_g1_decompress(const kzg_ctx* ctx, const uint8_t* in48, uint64_t n, uint8_t* out_affine96, int32_t* status) {
  const std::vector<Share> shares = shares_of(ctx, n);
  return run_on_helpers((uint32_t)shares.size(), [&](uint32_t j) -> int32_t {
    const Share& sh = shares[j];
    return g1_decompress_single(member_of(ctx, sh.member), in48 + sh.first * 48, sh.count, out_affine96 + sh.first * 96, status + sh.first);
  });
}

int32_t multi_evaluate_blobs(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* z32, uint64_t n, uint8_t* out_y32, int32_t* status) {
  const std::vector<Share> shares = shares_of(ctx, n);
  return run_on_helpers((uint32_t)shares.size(), [&](uint32_t j) -> int32_t {
    const Share& sh = shares[j];
    return evaluate_blobs_single(member_of(ctx, sh.member), blobs + sh.first * (size_t)KZG_BYTES_PER_BLOB, z32 + sh.first * 32, sh.count, out_y32 + sh.first * 32,
                                 status + sh.first);
  });
}

// Setup::verify_proof (src/kzg/setup.rs:96-113) is one item: any member serves it
int32_t multi_verify_proof(const kzg_ctx* ctx, const uint8_t* proof48, const uint8_t* commitment48, const uint8_t* z32, const uint8_t* y32, int32_t* ok) {
  const uint32_t S = 1u + (uint32_t)ctx->peers.size();
  return verify_proof_single(member_of(ctx, ctx->rr.fetch_add(1u, std::memory_order_relaxed) % S), proof48, commitment48, z32, y32, ok);
}

// Setup::verify_blob_proof_batch (src/kzg/setup.rs:247-275) over the members.  With one share this is exactly the
// single-device call (one root seeds the challenge); with several the challenge is seeded by all their roots, so r differs
// from the single-device call's while the boolean and the first-error code are the same.
int32_t multi_verify_batch(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* commitments48, const uint8_t* proofs48, uint64_t n, int32_t* ok) {
  *ok = 0;
  const std::vector<Share> shares = shares_of(ctx, n);
  const uint32_t W = (uint32_t)shares.size();
  if (W == 1) return verify_batch_host_single(member_of(ctx, shares[0].member), blobs, commitments48, proofs48, n, ok);
  std::vector<uint8_t> roots(32 * (size_t)W), partials(192 * (size_t)W);
  std::vector<int32_t> err6(6 * (size_t)W);
  std::vector<kzg_verify_session*> sessions(W, nullptr);
  auto release = [&]() {
    for (kzg_verify_session* s : sessions)
      if (s) kzg_verify_session_destroy(s);
  };
  int32_t rc = run_on_helpers(W, [&](uint32_t j) -> int32_t {
    const Share& sh = shares[j];
    const kzg_ctx* m = member_of(ctx, sh.member);
    if (hipSetDevice(m->device) != hipSuccess) return fail(KZG_FAIL_HIP, "hipSetDevice failed");
    return verify_phase1_host(m, blobs + sh.first * (size_t)KZG_BYTES_PER_BLOB, commitments48 + sh.first * 48, proofs48 + sh.first * 48, sh.count,
                              roots.data() + 32 * (size_t)j, err6.data() + 6 * (size_t)j, &sessions[j]);
  });
  if (rc) {
    const ErrorSnapshot keep = error_snapshot();
    release();
    error_publish(keep);
    return rc;
  }
  const int32_t code = merged_first_error(shares, err6.data());
  if (code) {
    release();
    return code;
  }
  rc = run_on_helpers(W, [&](uint32_t j) -> int32_t {
    return kzg_verify_phase2_dev(sessions[j], roots.data(), W, shares[j].first, n, partials.data() + 192 * (size_t)j);
  });
  {
    const ErrorSnapshot keep = error_snapshot();
    release();
    error_publish(keep);
  }
  if (rc) return rc;
  return kzg_verify_batch_finish(ctx, partials.data(), W, ok);
}

// ---- device-resident sharded calls (include/kateth_amd.h: kzg_*_group_dev) -------------------------------------------------
// Member k's share is resident on member k's GPU.  Commitments and proofs only ENQUEUE (like the *_dev calls), one pooled host
// thread per member so that the members' launches go out side by side; nothing is gathered -- results stay where they were
// computed.  Batch verification: engine_verify.hip (verify_group_dev).
namespace {
template <class Call>
int32_t group_enqueue(const kzg_ctx* ctx, const uint64_t* n_local, Call&& call) {
  const uint32_t S = 1u + (uint32_t)ctx->peers.size();
  std::vector<uint32_t> busy;
  for (uint32_t k = 0; k < S; k++)
    if (n_local[k]) busy.push_back(k);
  return run_on_helpers((uint32_t)busy.size(), [&](uint32_t j) -> int32_t { return call(busy[j], member_of(ctx, busy[j])); });
}
}  // namespace

extern "C" int32_t kzg_blob_to_commitment_batch_group_dev(const kzg_ctx* ctx, const void* const* d_blobs, const uint64_t* n_local, void* const* d_out48,
                                                          void* const* d_status, void* const* hip_streams) try {
  if (!ctx || !d_blobs || !n_local || !d_out48 || !d_status) return fail(KZG_FAIL_ARGUMENT, "null argument");
  return group_enqueue(ctx, n_local, [&](uint32_t k, const kzg_ctx* m) -> int32_t {
    return kzg_blob_to_commitment_batch_dev(m, d_blobs[k], n_local[k], d_out48[k], d_status[k], hip_streams ? hip_streams[k] : nullptr);
  });
} catch (...) {
  return abi_exception();
}

extern "C" int32_t kzg_compute_blob_proof_batch_group_dev(const kzg_ctx* ctx, const void* const* d_blobs, const void* const* d_commitments48,
                                                          const uint64_t* n_local, void* const* d_out48, void* const* d_status, void* const* hip_streams) try {
  if (!ctx || !d_blobs || !d_commitments48 || !n_local || !d_out48 || !d_status) return fail(KZG_FAIL_ARGUMENT, "null argument");
  return group_enqueue(ctx, n_local, [&](uint32_t k, const kzg_ctx* m) -> int32_t {
    return kzg_compute_blob_proof_batch_dev(m, d_blobs[k], d_commitments48[k], n_local[k], d_out48[k], d_status[k], hip_streams ? hip_streams[k] : nullptr);
  });
} catch (...) {
  return abi_exception();
}

extern "C" int32_t kzg_verify_blob_proof_batch_group_dev(const kzg_ctx* ctx, const void* const* d_blobs, const void* const* d_commitments48,
                                                         const void* const* d_proofs48, const uint64_t* n_local, int32_t* ok, void* const* hip_streams) try {
  if (!ctx || !ok || !d_blobs || !d_commitments48 || !d_proofs48 || !n_local) return fail(KZG_FAIL_ARGUMENT, "null argument");
  *ok = 0;
  const uint32_t S = 1u + (uint32_t)ctx->peers.size();
  std::vector<GroupDevShare> shares;
  uint64_t total = 0;
  for (uint32_t k = 0; k < S; k++) {
    if (n_local[k] == 0) continue;
    if (!d_blobs[k] || !d_commitments48[k] || !d_proofs48[k]) return fail(KZG_FAIL_ARGUMENT, "null device pointer for a member with items");
    shares.push_back(GroupDevShare{member_of(ctx, k), (const uint8_t*)d_blobs[k], (const uint8_t*)d_commitments48[k], (const uint8_t*)d_proofs48[k], total, n_local[k],
                                   hip_streams ? (hipStream_t)hip_streams[k] : nullptr});
    total += n_local[k];
  }
  return verify_group_dev(ctx, shares, total, ok);
} catch (...) {
  return abi_exception();
}
