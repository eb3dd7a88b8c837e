// C++ host-side mirror of kateth's public API over the C ABI (include/kateth_amd.h).
//
// The reference is a Rust crate; no Rust toolchain exists in the build image, so
// the compiled-language mirror of `kateth::kzg::Setup<4096, 65>` is C++.  Names,
// argument meaning and error behaviour follow the reference:
//
//   Setup::load(g1, g2)                  <- Setup::load_json after parsing   src/kzg/setup.rs:46-82
//   blob_to_commitment(blob)             <- src/kzg/setup.rs:167-171 (+ compress, src/bls.rs:491-503)
//   blob_proof(blob, commitment)         <- src/kzg/setup.rs:177-183
//   proof(blob, z)                       <- src/kzg/setup.rs:185-194
//   verify_proof(proof, commitment, z, y)<- src/kzg/setup.rs:96-113
//   verify_blob_proof(blob, c, p)        <- src/kzg/setup.rs:208-221
//   verify_blob_proof_batch(blobs,cs,ps) <- src/kzg/setup.rs:247-275
//
// `Result<T, E>` becomes a return value + thrown `kateth::Error` (the Err arm);
// the reference's `assert_eq!` on batch lengths (src/kzg/setup.rs:256-257) becomes
// std::logic_error.  Header-only; link with libkateth_amd.so and the HIP runtime.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/kateth_amd.h"

namespace kateth {

using Bytes32 = std::array<uint8_t, 32>;  // src/kzg/mod.rs:12
using Bytes48 = std::array<uint8_t, 48>;  // src/kzg/mod.rs:13

// kzg::Error { Blob(blob::Error), Bls(bls::Error) }  (src/kzg/mod.rs:15-31)
enum class ErrorKind : int32_t {
  BlobInvalidLen = KZG_ERR_BLOB_INVALID_LEN,                     // src/blob.rs:9
  BlobInvalidFieldElement = KZG_ERR_BLOB_INVALID_FIELD_ELEMENT,  // src/blob.rs:8
  ECGroupInvalidEncoding = KZG_ERR_EC_INVALID_ENCODING,          // src/bls.rs:29
  ECGroupNotOnCurve = KZG_ERR_EC_NOT_ON_CURVE,                   // src/bls.rs:31
  ECGroupNotInGroup = KZG_ERR_EC_NOT_IN_GROUP,                   // src/bls.rs:30
  FiniteFieldInvalidEncoding = KZG_ERR_FF_INVALID_ENCODING,      // src/bls.rs:23
  FiniteFieldNotInFiniteField = KZG_ERR_FF_NOT_IN_FIELD,         // src/bls.rs:24
};

class Error : public std::runtime_error {
 public:
  explicit Error(ErrorKind k) : std::runtime_error("kateth::Error(" + std::to_string(static_cast<int>(k)) + ")"), kind(k) {}
  ErrorKind kind;
};

// negative return from the engine: HIP / device / argument failure (no CPU fallback exists)
class EngineFailure : public std::runtime_error {
 public:
  EngineFailure(const char* what, int32_t rc) : std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + kzg_last_error()), code(rc) {}
  int32_t code;
};

template <size_t G1 = 4096, size_t G2 = 65>
class Setup {
  static_assert(G1 == KZG_SETUP_G1_POINTS && G2 == KZG_SETUP_G2_POINTS, "the engine is built for Setup<4096, 65> (benches/kzg.rs:12)");

 public:
  static constexpr size_t BLOB_BYTES = KZG_BYTES_PER_BLOB;  // Blob::<4096>::BYTES

  // g1_lagrange: 4096 x 48 B, g2_monomial: 65 x 96 B, in file order
  static Setup load(const uint8_t* g1_lagrange, const uint8_t* g2_monomial, int device = 0, int window_bits = 0) {
    kzg_config cfg{device, window_bits, 0, 0};
    kzg_ctx* ctx = nullptr;
    int32_t rc = kzg_ctx_create(g1_lagrange, g2_monomial, &cfg, &ctx);
    if (rc != 0) throw EngineFailure("kzg_ctx_create", rc);  // includes LoadSetupError::Bls (-4 / -5)
    return Setup(ctx);
  }

  Bytes48 blob_to_commitment(const uint8_t* blob, size_t len) const {
    if (len != BLOB_BYTES) throw Error(ErrorKind::BlobInvalidLen);  // src/blob.rs:27-29
    Bytes48 out{};
    int32_t status = 0;
    check(kzg_blob_to_commitment_batch(ctx_.get(), blob, 1, out.data(), &status), "kzg_blob_to_commitment_batch");
    if (status) throw Error(static_cast<ErrorKind>(status));
    return out;
  }

  Bytes48 blob_proof(const uint8_t* blob, size_t len, const Bytes48& commitment) const {
    if (len != BLOB_BYTES) throw Error(ErrorKind::BlobInvalidLen);
    Bytes48 out{};
    int32_t status = 0;
    check(kzg_compute_blob_proof_batch(ctx_.get(), blob, commitment.data(), 1, out.data(), &status), "kzg_compute_blob_proof_batch");
    if (status) throw Error(static_cast<ErrorKind>(status));
    return out;
  }

  // (proof, y)
  std::pair<Bytes48, Bytes32> proof(const uint8_t* blob, size_t len, const Bytes32& point) const {
    if (len != BLOB_BYTES) throw Error(ErrorKind::BlobInvalidLen);
    Bytes48 pi{};
    Bytes32 y{};
    int32_t status = 0;
    check(kzg_compute_proof_batch(ctx_.get(), blob, point.data(), 1, pi.data(), y.data(), &status), "kzg_compute_proof_batch");
    if (status) throw Error(static_cast<ErrorKind>(status));
    return {pi, y};
  }

  bool verify_proof(const Bytes48& proof, const Bytes48& commitment, const Bytes32& point, const Bytes32& eval) const {
    int32_t ok = 0;
    int32_t rc = kzg_verify_proof(ctx_.get(), proof.data(), commitment.data(), point.data(), eval.data(), &ok);
    return finish(rc, ok, "kzg_verify_proof");
  }

  bool verify_blob_proof(const uint8_t* blob, size_t len, const Bytes48& commitment, const Bytes48& proof) const {
    if (len != BLOB_BYTES) throw Error(ErrorKind::BlobInvalidLen);
    int32_t ok = 0;
    int32_t rc = kzg_verify_blob_proof(ctx_.get(), blob, commitment.data(), proof.data(), &ok);
    return finish(rc, ok, "kzg_verify_blob_proof");
  }

  // blobs: n pointers to 131072-byte buffers (lengths in blob_lens)
  bool verify_blob_proof_batch(const std::vector<const uint8_t*>& blobs, const std::vector<size_t>& blob_lens,
                               const std::vector<Bytes48>& commitments, const std::vector<Bytes48>& proofs) const {
    if (blobs.size() != commitments.size() || commitments.size() != proofs.size() || blobs.size() != blob_lens.size())
      throw std::logic_error("assertion `left == right` failed");  // src/kzg/setup.rs:256-257 panics
    const size_t n = blobs.size();
    for (size_t i = 0; i < n; i++)
      if (blob_lens[i] != BLOB_BYTES) throw Error(ErrorKind::BlobInvalidLen);
    std::vector<uint8_t> flat(n * BLOB_BYTES), cs(n * 48), ps(n * 48);
    for (size_t i = 0; i < n; i++) {
      std::copy(blobs[i], blobs[i] + BLOB_BYTES, flat.begin() + i * BLOB_BYTES);
      std::copy(commitments[i].begin(), commitments[i].end(), cs.begin() + i * 48);
      std::copy(proofs[i].begin(), proofs[i].end(), ps.begin() + i * 48);
    }
    int32_t ok = 0;
    int32_t rc = kzg_verify_blob_proof_batch(ctx_.get(), flat.data(), cs.data(), ps.data(), n, &ok);
    return finish(rc, ok, "kzg_verify_blob_proof_batch");
  }

  // batch forms (contiguous buffers) for callers that already hold many blobs
  void blob_to_commitment_batch(const uint8_t* blobs, size_t n, uint8_t* out48, int32_t* status) const {
    check(kzg_blob_to_commitment_batch(ctx_.get(), blobs, n, out48, status), "kzg_blob_to_commitment_batch");
  }
  void blob_proof_batch(const uint8_t* blobs, const uint8_t* commitments48, size_t n, uint8_t* out48, int32_t* status) const {
    check(kzg_compute_blob_proof_batch(ctx_.get(), blobs, commitments48, n, out48, status), "kzg_compute_blob_proof_batch");
  }

  const kzg_ctx* raw() const { return ctx_.get(); }

 private:
  struct Deleter {
    void operator()(kzg_ctx* c) const { kzg_ctx_destroy(c); }
  };
  explicit Setup(kzg_ctx* c) : ctx_(c, Deleter()) {}
  static void check(int32_t rc, const char* what) {
    if (rc < 0) throw EngineFailure(what, rc);
  }
  static bool finish(int32_t rc, int32_t ok, const char* what) {
    if (rc < 0) throw EngineFailure(what, rc);
    if (rc > 0) throw Error(static_cast<ErrorKind>(rc));
    return ok != 0;
  }
  std::shared_ptr<kzg_ctx> ctx_;  // Arc<Setup>: cheap to share between threads

 public:
  Setup(const Setup&) = default;
  Setup(Setup&&) noexcept = default;
};

}  // namespace kateth
