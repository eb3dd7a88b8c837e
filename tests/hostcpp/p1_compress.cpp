// kateth::P1::compress and kateth::Blob (host mirror, no GPU needed): argv[1] = 96-byte affine image in hex -> 48-byte encoding in hex;
// with "blob" as argv[1]: a Blob::random round trip through from_slice.
#include <random>
#include <cstring>
#include "../../kateth_amd/host/kateth.hpp"
#include <cstdio>
int main(int argc,char**argv){ if (argc > 1 && !strcmp(argv[1], "blob")) { std::mt19937_64 g(1); auto b = kateth::Blob::random(g); auto c = kateth::Blob::from_slice(b.to_bytes().data(), b.to_bytes().size()); std::vector<uint8_t> bad = b.to_bytes(); memset(bad.data() + 64, 0xff, 32); try { kateth::Blob::from_slice(bad.data(), bad.size()); return 3; } catch (const kateth::Error& e) { if (e.kind != kateth::ErrorKind::BlobInvalidFieldElement) return 4; } try { kateth::Blob::from_slice(bad.data(), bad.size() - 1); return 5; } catch (const kateth::Error& e) { if (e.kind != kateth::ErrorKind::BlobInvalidLen) return 6; } printf("%zu\n", c.to_bytes().size()); return 0; } kateth::P1 p; for(int i=0;i<96;i++){unsigned v; sscanf(argv[1]+2*i,"%2x",&v); p.affine[i]=v;} auto c=p.compress(); for(auto b:c) printf("%02x",b); printf("\n"); }
