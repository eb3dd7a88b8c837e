// Compile-and-link check of the C++ host mirror (kateth_amd/host/kateth.hpp) against the C ABI.
// On a machine without a GPU it must fail loudly with KZG_FAIL_NO_DEVICE; with one it runs a tiny round trip.
#include <cstdio>
#include <random>
#include <vector>

#include "../../kateth_amd/host/kateth.hpp"

int main(int argc, char** argv) {
  std::vector<uint8_t> g1(4096 * 48), g2(65 * 96);
  if (argc >= 3) {  // raw setup bytes supplied by the test
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(g1.data(), 1, g1.size(), f) != g1.size()) return 10;
    fclose(f);
    f = fopen(argv[2], "rb");
    if (!f || fread(g2.data(), 1, g2.size(), f) != g2.size()) return 11;
    fclose(f);
  }
  try {
    auto setup = kateth::Setup<4096, 65>::load(g1.data(), g2.data(), 0, 8);
    std::vector<uint8_t> blob(kateth::Setup<>::BLOB_BYTES, 0);
    for (size_t i = 0; i < 4096; i++) blob[32 * i + 31] = 1;  // all-ones blob -> commitment = G1 generator
    kateth::Bytes48 c = setup.blob_to_commitment(blob.data(), blob.size());
    kateth::Bytes48 p = setup.blob_proof(blob.data(), blob.size(), c);
    bool ok = setup.verify_blob_proof(blob.data(), blob.size(), c, p);
    std::printf("commitment[0]=%02x proof[0]=%02x verify=%d\n", c[0], p[0], (int)ok);
    // Blob::random -> commit (point and bytes) -> prove -> verify, the input pipeline of benches/kzg.rs:17-33
    std::mt19937_64 gen(4844);
    kateth::Blob rb = kateth::Blob::random(gen);
    kateth::Blob::from_slice(rb.to_bytes().data(), rb.to_bytes().size());  // every element canonical
    kateth::Bytes48 rc = setup.blob_to_commitment(rb.to_bytes().data(), kateth::Blob::BYTES);
    kateth::P1 rcp = setup.blob_to_commitment_point(rb.to_bytes().data(), kateth::Blob::BYTES);
    kateth::Bytes48 rp = setup.blob_proof(rb.to_bytes().data(), kateth::Blob::BYTES, rc);
    const bool ok2 = setup.verify_blob_proof(rb.to_bytes().data(), kateth::Blob::BYTES, rc, rp) && !rcp.is_inf() && rcp.compress() == rc;  // benches/kzg.rs:24-26
    std::printf("random blob: commitment[0]=%02x affine[0]=%02x verify=%d\n", rc[0], rcp.affine[0], (int)ok2);
    if (!ok2) return 6;
    try {
      setup.blob_to_commitment(blob.data(), blob.size() - 1);
      return 3;
    } catch (const kateth::Error& e) {
      if (e.kind != kateth::ErrorKind::BlobInvalidLen) return 4;
    }
    // first-error order of verify_blob_proof_batch (src/kzg/setup.rs:259-262: the FIRST failing blob decides): blob 0 non-canonical
    // and blob 1 short is InvalidFieldElement; blob 0 fine and blob 1 short is InvalidLen
    {
      std::vector<uint8_t> nc(rb.to_bytes());
      for (int k = 0; k < 32; k++) nc[k] = 0xff;  // element 0 >= r
      for (int variant = 0; variant < 2; variant++) {
        std::vector<const uint8_t*> bl = {variant == 0 ? nc.data() : rb.to_bytes().data(), rb.to_bytes().data()};
        std::vector<size_t> lens = {kateth::Blob::BYTES, kateth::Blob::BYTES - 1};
        try {
          setup.verify_blob_proof_batch(bl, lens, {rc, rc}, {rp, rp});
          return 10;
        } catch (const kateth::Error& e) {
          if (e.kind != (variant == 0 ? kateth::ErrorKind::BlobInvalidFieldElement : kateth::ErrorKind::BlobInvalidLen)) return 11;
        }
      }
    }
    // the same over a GROUP context (the device listed twice: two members on the card): every batch is sharded behind the same methods
    auto group = kateth::Setup<4096, 65>::load_multi(g1.data(), g2.data(), {0, 0}, 8);
    if (group.members() != 2 || setup.members() != 1) return 7;
    std::vector<uint8_t> three(3 * kateth::Blob::BYTES), out3(3 * 48), ref3(3 * 48);
    std::vector<int32_t> st3(3), st3r(3);
    for (int k = 0; k < 3; k++) std::copy(rb.to_bytes().begin(), rb.to_bytes().end(), three.begin() + k * kateth::Blob::BYTES);
    three[kateth::Blob::BYTES + 31] ^= 1;  // the middle blob differs
    group.blob_to_commitment_batch(three.data(), 3, out3.data(), st3.data());
    setup.blob_to_commitment_batch(three.data(), 3, ref3.data(), st3r.data());
    if (out3 != ref3 || st3 != st3r || !std::equal(out3.begin(), out3.begin() + 48, rc.begin())) return 8;
    if (!group.verify_blob_proof(rb.to_bytes().data(), kateth::Blob::BYTES, rc, rp)) return 9;
    return (ok && c[0] == 0x97 && p[0] == 0xc0) ? 0 : 2;
  } catch (const kateth::EngineFailure& e) {
    std::printf("engine failure %d: %s\n", e.code, e.what());
    return e.code == KZG_FAIL_NO_DEVICE ? 42 : 5;
  }
}
