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
//   decompress(bytes48) -> P1            <- P1::decompress, src/bls.rs:505-531
//   verify_blob_proof(blob, c, p)        <- src/kzg/setup.rs:208-221
//   verify_blob_proof_batch(blobs,cs,ps) <- src/kzg/setup.rs:247-275
//   Blob::{from_slice,to_bytes,random}   <- src/blob.rs:26-46,66-76
//   P1 (+ compress)                      <- Commitment = Proof = P1, src/kzg/mod.rs:9-10; Compress, src/bls.rs:491-503
//
// The byte-returning methods give the 48-byte encodings; the *_point methods return `P1`, the type the reference's
// producers return (src/kzg/setup.rs:167,177,185), so that call sites shaped like benches/kzg.rs:24-32
// (`kzg.blob_to_commitment(blob).unwrap().compress(&mut bytes)`) carry over.
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

// `bls::P1` as handed over by the engine: the 96-byte blst_p1_affine image (x || y, 6 x u64 little-endian limbs each, of
// the 2^384-Montgomery residues; all zero = infinity) -- what a Rust caller passes to blst_p1_from_affine.
struct P1 {
  static constexpr size_t COMPRESSED = KZG_BYTES_PER_G1;
  std::array<uint8_t, 96> affine{};
  bool is_inf() const {
    for (uint8_t b : affine)
      if (b) return false;
    return true;
  }
  bool operator==(const P1& o) const { return affine == o.affine; }
  // `Compress::compress` (src/bls.rs:491-503; a blst CPU call in the reference, made by the caller on the returned point):
  // a change of encoding of the already-normalised point -- leave the Montgomery domain, big-endian x, flag bits.
  Bytes48 compress() const {
    Bytes48 out{};
    if (is_inf()) {
      out[0] = 0xC0;
      return out;
    }
    static constexpr uint64_t P[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                                      0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
    static constexpr uint64_t HALF[6] = {0xdcff7fffffffd555ull, 0x0f55ffff58a9ffffull, 0xb39869507b587b12ull,
                                         0xb23ba5c279c2895full, 0x258dd3db21a5d66bull, 0x0d0088f51cbff34dull};  // (p - 1) / 2
    constexpr uint64_t N0 = 0x89f3fffcfffcfffdull;                                                            // -p^-1 mod 2^64
    auto from_mont = [&](const uint8_t* in, uint64_t* r) {  // REDC(a) = a * 2^-384 mod p
      uint64_t t[7] = {0};
      for (int i = 0; i < 6; i++)
        for (int b = 0; b < 8; b++) t[i] |= (uint64_t)in[8 * i + b] << (8 * b);
      for (int i = 0; i < 6; i++) {
        const uint64_t m = t[0] * N0;
        unsigned __int128 c = (unsigned __int128)m * P[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 6; j++) {
          c += (unsigned __int128)m * P[j] + t[j];
          t[j - 1] = (uint64_t)c;
          c >>= 64;
        }
        c += t[6];
        t[5] = (uint64_t)c;
        t[6] = (uint64_t)(c >> 64);
      }
      bool ge = t[6] != 0;
      if (!ge) {
        ge = true;
        for (int j = 5; j >= 0; j--)
          if (t[j] != P[j]) {
            ge = t[j] > P[j];
            break;
          }
      }
      if (ge) {
        unsigned __int128 bw = 0;
        for (int j = 0; j < 6; j++) {
          const unsigned __int128 d = (unsigned __int128)t[j] - P[j] - bw;
          t[j] = (uint64_t)d;
          bw = (d >> 64) & 1;
        }
      }
      for (int j = 0; j < 6; j++) r[j] = t[j];
    };
    uint64_t x[6], y[6];
    from_mont(affine.data(), x);
    from_mont(affine.data() + 48, y);
    for (int j = 0; j < 6; j++)
      for (int b = 0; b < 8; b++) out[47 - (8 * j + b)] = (uint8_t)(x[j] >> (8 * b));
    bool larger = false;
    for (int j = 5; j >= 0; j--)
      if (y[j] != HALF[j]) {
        larger = y[j] > HALF[j];
        break;
      }
    out[0] |= 0x80 | (larger ? 0x20 : 0);
    return out;
  }
};

namespace detail {
// SHA-256 for Blob::random (Fr::hash_to, src/bls.rs:189-205); input generation only, nothing on the hot path
inline void sha256(uint8_t out[32], const uint8_t* msg, size_t len) {
  static const uint32_t K[64] = {
      0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74,
      0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d,
      0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e,
      0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5,
      0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  auto rotr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
  std::vector<uint8_t> m(msg, msg + len);
  m.push_back(0x80);
  while (m.size() % 64 != 56) m.push_back(0);
  for (int k = 7; k >= 0; k--) m.push_back((uint8_t)((uint64_t)len * 8 >> (8 * k)));
  for (size_t off = 0; off < m.size(); off += 64) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = (uint32_t)m[off + 4 * i] << 24 | (uint32_t)m[off + 4 * i + 1] << 16 | (uint32_t)m[off + 4 * i + 2] << 8 | m[off + 4 * i + 3];
    for (int i = 16; i < 64; i++)
      w[i] = w[i - 16] + (rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10));
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
      const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
      const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  for (int i = 0; i < 8; i++)
    for (int k = 0; k < 4; k++) out[4 * i + k] = (uint8_t)(h[i] >> (24 - 8 * k));
}
// the Fr modulus, big-endian
static constexpr uint8_t FR_MODULUS_BE[32] = {0x73, 0xed, 0xa7, 0x53, 0x29, 0x9d, 0x7d, 0x48, 0x33, 0x39, 0xd8, 0x08, 0x09, 0xa1, 0xd8, 0x05,
                                              0x53, 0xbd, 0xa4, 0x02, 0xff, 0xfe, 0x5b, 0xfe, 0xff, 0xff, 0xff, 0xff, 0x00, 0x00, 0x00, 0x01};
inline bool be32_less(const uint8_t* a, const uint8_t* b) {  // a < b as 256-bit big-endian integers
  for (int i = 0; i < 32; i++)
    if (a[i] != b[i]) return a[i] < b[i];
  return false;
}
inline void be32_sub_modulus(uint8_t* a) {
  int borrow = 0;
  for (int i = 31; i >= 0; i--) {
    int v = (int)a[i] - FR_MODULUS_BE[i] - borrow;
    borrow = v < 0;
    a[i] = (uint8_t)(v + (borrow ? 256 : 0));
  }
}
}  // namespace detail

// `Blob<4096>` (src/blob.rs:18-76): a validated 131,072-byte blob; every computation on it happens in the engine.
class Blob {
 public:
  static constexpr size_t BYTES = KZG_BYTES_PER_BLOB;
  static Blob from_slice(const uint8_t* bytes, size_t len) {  // src/blob.rs:26-37
    if (len != BYTES) throw Error(ErrorKind::BlobInvalidLen);
    for (size_t i = 0; i < BYTES; i += 32)
      if (!detail::be32_less(bytes + i, detail::FR_MODULUS_BE)) throw Error(ErrorKind::BlobInvalidFieldElement);
    return Blob(std::vector<uint8_t>(bytes, bytes + len));
  }
  const std::vector<uint8_t>& to_bytes() const { return data_; }  // src/blob.rs:39-46
  // src/blob.rs:66-76: element = Fr::hash_to(512 bytes from `gen`) = SHA-256 mod r; Gen: any callable returning uint64_t (e.g. std::mt19937_64)
  template <class Gen>
  static Blob random(Gen& gen) {
    std::vector<uint8_t> data(BYTES);
    uint8_t buf[512];
    for (size_t i = 0; i < BYTES; i += 32) {
      for (size_t k = 0; k < 512; k += 8) {
        const uint64_t v = (uint64_t)gen();
        for (int b = 0; b < 8; b++) buf[k + b] = (uint8_t)(v >> (8 * b));
      }
      detail::sha256(&data[i], buf, 512);
      while (!detail::be32_less(&data[i], detail::FR_MODULUS_BE)) detail::be32_sub_modulus(&data[i]);  // 2^256 < 3 r: at most two
    }
    return Blob(std::move(data));
  }

 private:
  explicit Blob(std::vector<uint8_t> d) : data_(std::move(d)) {}
  std::vector<uint8_t> data_;
};

template <size_t G1 = 4096, size_t G2 = 65>
class Setup {
  static_assert(G1 == KZG_SETUP_G1_POINTS && G2 == KZG_SETUP_G2_POINTS, "the engine is built for Setup<4096, 65> (benches/kzg.rs:12)");

 public:
  static constexpr size_t BLOB_BYTES = KZG_BYTES_PER_BLOB;  // Blob::<4096>::BYTES

  // g1_lagrange: 4096 x 48 B, g2_monomial: 65 x 96 B, in file order
  // window_bits = 0 / plane_groups = 0: the engine picks the fastest table class within its default budget (the 96-GiB table)
  // that the device has room for; flags: KZG_CFG_TABLE_MAX lifts the budget, KZG_CFG_BUILD_ASYNC returns on a first-use table
  static Setup load(const uint8_t* g1_lagrange, const uint8_t* g2_monomial, int device = 0, int window_bits = 0, int plane_groups = 0, int flags = 0) {
    kzg_config cfg = KZG_CONFIG_INIT;
    cfg.device = device;
    cfg.window_bits = window_bits;
    cfg.flags = flags;
    cfg.plane_groups = plane_groups;
    kzg_ctx* ctx = nullptr;
    int32_t rc = kzg_ctx_create(g1_lagrange, g2_monomial, &cfg, &ctx);
    if (rc != 0) throw EngineFailure("kzg_ctx_create", rc);  // includes LoadSetupError::Bls (-4 / -5)
    return Setup(ctx);
  }
  // The same over several GPUs of the node (an empty list = every visible device): a GROUP context whose methods shard
  // every host-buffer batch over the members (include/kateth_amd.h, kzg_config.devices); single items go to a rotating member.
  static Setup load_multi(const uint8_t* g1_lagrange, const uint8_t* g2_monomial, const std::vector<int32_t>& devices = {}, int window_bits = 0,
                          int plane_groups = 0, int flags = 0) {
    kzg_config cfg = KZG_CONFIG_INIT;
    cfg.window_bits = window_bits;
    cfg.flags = flags;
    cfg.plane_groups = plane_groups;
    kzg_ctx* ctx = nullptr;
    int32_t rc = kzg_ctx_create_multi(g1_lagrange, g2_monomial, devices.empty() ? nullptr : devices.data(),
                                      devices.empty() ? KZG_ALL_DEVICES : (uint32_t)devices.size(), &cfg, &ctx);
    if (rc != 0) throw EngineFailure("kzg_ctx_create_multi", rc);
    return Setup(ctx);
  }
  uint32_t members() const { return kzg_ctx_members(ctx_.get()); }
  void wait_ready() const { check(kzg_ctx_wait_ready(ctx_.get()), "kzg_ctx_wait_ready"); }

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

  // ---- the same producers with the reference's return type (src/kzg/setup.rs:167,177,185) -------------------------
  P1 blob_to_commitment_point(const uint8_t* blob, size_t len) const {
    if (len != BLOB_BYTES) throw Error(ErrorKind::BlobInvalidLen);
    P1 out;
    int32_t status = 0;
    check(kzg_blob_to_commitment_batch_affine(ctx_.get(), blob, 1, out.affine.data(), &status), "kzg_blob_to_commitment_batch_affine");
    if (status) throw Error(static_cast<ErrorKind>(status));
    return out;
  }
  P1 blob_proof_point(const uint8_t* blob, size_t len, const Bytes48& commitment) const {
    if (len != BLOB_BYTES) throw Error(ErrorKind::BlobInvalidLen);
    P1 out;
    int32_t status = 0;
    check(kzg_compute_blob_proof_batch_affine(ctx_.get(), blob, commitment.data(), 1, out.affine.data(), &status), "kzg_compute_blob_proof_batch_affine");
    if (status) throw Error(static_cast<ErrorKind>(status));
    return out;
  }
  std::pair<P1, Bytes32> proof_point(const uint8_t* blob, size_t len, const Bytes32& point) const {
    if (len != BLOB_BYTES) throw Error(ErrorKind::BlobInvalidLen);
    P1 pi;
    Bytes32 y{};
    int32_t status = 0;
    check(kzg_compute_proof_batch_affine(ctx_.get(), blob, point.data(), 1, pi.affine.data(), y.data(), &status), "kzg_compute_proof_batch_affine");
    if (status) throw Error(static_cast<ErrorKind>(status));
    return {pi, y};
  }

  // `P1::decompress` (the crate's `Decompress` trait on Commitment / Proof, src/bls.rs:505-531): decode + on-curve + subgroup
  // check of one point on the engine; throws the reference's ECGroupError for a rejected encoding
  P1 decompress(const Bytes48& in) const {
    P1 out;
    int32_t status = 0;
    int32_t rc = kzg_g1_decompress_batch(ctx_.get(), in.data(), 1, out.affine.data(), &status);
    if (rc < 0) throw EngineFailure("kzg_g1_decompress_batch", rc);
    if (status) throw Error(static_cast<ErrorKind>(status));
    return out;
  }

  // `Polynomial::evaluate` (src/kzg/poly.rs:10-33) of one blob at one point, through the verification path's evaluation kernel
  Bytes32 evaluate(const uint8_t* blob, size_t len, const Bytes32& point) const {
    if (len != BLOB_BYTES) throw Error(ErrorKind::BlobInvalidLen);
    Bytes32 y{};
    int32_t status = 0;
    check(kzg_evaluate_blobs(ctx_.get(), blob, point.data(), 1, y.data(), &status), "kzg_evaluate_blobs");
    if (status) throw Error(static_cast<ErrorKind>(status));
    return y;
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
    // the reference's `collect` stops at the FIRST blob that fails, whatever its error (src/kzg/setup.rs:259-262): a short blob
    // never reaches the engine, so the blobs before it get Blob::from_slice's element check (src/blob.rs:32-34) here
    for (size_t i = 0; i < n; i++)
      if (blob_lens[i] != BLOB_BYTES) {
        for (size_t j = 0; j < i; j++)
          if (has_noncanonical_element(blobs[j])) throw Error(ErrorKind::BlobInvalidFieldElement);
        throw Error(ErrorKind::BlobInvalidLen);
      }
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

  // some 32-byte big-endian element of a full-length blob is >= r (host side, error path only)
  static bool has_noncanonical_element(const uint8_t* blob) {
    static const uint8_t r_be[32] = {0x73, 0xed, 0xa7, 0x53, 0x29, 0x9d, 0x7d, 0x48, 0x33, 0x39, 0xd8, 0x08, 0x09, 0xa1, 0xd8, 0x05,
                                     0x53, 0xbd, 0xa4, 0x02, 0xff, 0xfe, 0x5b, 0xfe, 0xff, 0xff, 0xff, 0xff, 0x00, 0x00, 0x00, 0x01};
    for (size_t e = 0; e < KZG_FIELD_ELEMENTS_PER_BLOB; e++) {
      const uint8_t* p = blob + 32 * e;
      int cmp = 0;
      for (int k = 0; k < 32 && cmp == 0; k++) cmp = (p[k] > r_be[k]) - (p[k] < r_be[k]);
      if (cmp >= 0) return true;
    }
    return false;
  }

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
