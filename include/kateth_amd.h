/*
 * kateth_amd -- MI355X (gfx950) EIP-4844 KZG engine: C ABI.
 *
 * This is the drop-in boundary for kateth's hot path.  kateth has no FFI of its
 * own below its public Rust API other than the per-operation blst calls
 * (src/bls.rs:8-19), which are too fine-grained to put a GPU behind, so the
 * entry points here are batch-level and sit directly under the crate's public
 * methods; each one names the reference function it
 * replaces.  Byte layouts are exactly the reference's wire formats: a blob is
 * 4096 x 32-byte big-endian field elements (src/blob.rs:23-37), a commitment or
 * proof is a 48-byte ZCash-compressed G1 point (src/bls.rs:491-531), a field
 * element is 32 bytes big-endian (src/bls.rs:130-149).
 *
 * Plain C: pointers and sizes only.  Nothing unwinds across this boundary.
 * A kzg_ctx is immutable after creation and may be used from several host
 * threads at once, matching `&self` + `Arc<Setup>` in the reference
 * (src/kzg/setup.rs:323): commitment and proof calls hold an internal lock only
 * while they enqueue and take one of the context's three GPU workspaces (the lowest
 * one whose last user has completed or ran on the call's own stream), so calls
 * enqueued on several streams run side by side; every verification call takes its
 * own pooled session (device scratch + streams) and runs beside the others; the
 * host-buffer calls additionally serialise on the staging arena.  A context
 * created with kzg_config.devices / ndev spans several GPUs (below).
 *
 * Streams and hardware queues: the HIP runtime multiplexes all streams of a
 * process onto GPU_MAX_HW_QUEUES hardware queues (4 unless set) and streams
 * that share a queue run one after the other.  The variable is read when the
 * PROCESS first touches HIP, so it is the host program's to set: the library
 * never changes the environment.  kzg_recommended_env() returns the setting
 * the measured configurations ran with ("GPU_MAX_HW_QUEUES=16"); with
 * KATETH_AMD_TRACE set, kzg_ctx_create says once when it is missing or below
 * 8.  Results never depend on it -- only how many
 * calls kept in flight on different streams really overlap.
 *
 * Return value of every call: 0 on success, a positive KZG_ERR_* code when an
 * input is rejected the way the reference returns Err, a negative KZG_FAIL_*
 * code for a runtime (HIP / allocation / argument) failure.
 */
#ifndef KATETH_AMD_H
#define KATETH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KZG_FIELD_ELEMENTS_PER_BLOB 4096
#define KZG_BYTES_PER_FIELD_ELEMENT 32
#define KZG_BYTES_PER_BLOB 131072 /* Blob::<4096>::BYTES, src/blob.rs:24 */
#define KZG_BYTES_PER_G1 48       /* P1::COMPRESSED, src/bls.rs:552 */
#define KZG_BYTES_PER_G2 96       /* P2::COMPRESSED, src/bls.rs:569 */
#define KZG_SETUP_G1_POINTS 4096  /* Setup<4096, 65>, benches/kzg.rs:12 */
#define KZG_SETUP_G2_POINTS 65

/* per-item / per-call rejection codes; they rebuild the reference's enums */
#define KZG_OK 0
#define KZG_ERR_BLOB_INVALID_LEN 1           /* blob::Error::InvalidLen            src/blob.rs:9   */
#define KZG_ERR_BLOB_INVALID_FIELD_ELEMENT 2 /* blob::Error::InvalidFieldElement   src/blob.rs:8   */
#define KZG_ERR_EC_INVALID_ENCODING 3        /* ECGroupError::InvalidEncoding      src/bls.rs:29  */
#define KZG_ERR_EC_NOT_ON_CURVE 4            /* ECGroupError::NotOnCurve           src/bls.rs:31  */
#define KZG_ERR_EC_NOT_IN_GROUP 5            /* ECGroupError::NotInGroup           src/bls.rs:30  */
#define KZG_ERR_FF_INVALID_ENCODING 6        /* FiniteFieldError::InvalidEncoding  src/bls.rs:23  */
#define KZG_ERR_FF_NOT_IN_FIELD 7            /* FiniteFieldError::NotInFiniteField src/bls.rs:24  */

#define KZG_FAIL_ARGUMENT (-1)
#define KZG_FAIL_HIP (-2)
#define KZG_FAIL_NO_DEVICE (-3)
#define KZG_FAIL_SETUP_G1 (-4) /* LoadSetupError::Bls on a g1_lagrange point, src/kzg/setup.rs:59-64 */
#define KZG_FAIL_SETUP_G2 (-5) /* LoadSetupError::Bls on a g2_monomial point, src/kzg/setup.rs:67-72 */
/* A setup the comb table cannot represent: some +-1 combination of a block of consecutive g1_lagrange points is the point
 * at infinity (repeated / opposite points).  The reference would load it; no ceremony output or set of independent points
 * is affected.  Rejected at creation instead of producing wrong commitments. */
#define KZG_FAIL_SETUP_UNSUPPORTED (-6)
/* The host side of a call failed: out of host memory, or a C++ exception nobody expected (kzg_last_error() has its text).
 * Every int32_t entry point catches what its body throws -- nothing unwinds into the caller's frames. */
#define KZG_FAIL_HOST (-7)

typedef struct kzg_ctx kzg_ctx;

/*
 * The fixed-base MSM table is a subset-sum comb (kateth_amd/csrc/msm_comb.cuh): the 4096 Lagrange points are cut into blocks
 * of t consecutive points, a table entry (96 B, affine) is one +-1 combination of a block, and the 256 bit planes of the
 * scalars are cut into G plane groups with a table each.  Table bytes = G * 64 * e * 96 with e entries per 64 points and
 * group; mixed additions per blob = 256 * (blocks per 64 points) * 64; a lane doubles its accumulator 256/G - 1 times.
 *
 *   class | blocks per 64 points | e        | default G | resident table   | additions per blob
 *   ------+----------------------+----------+-----------+------------------+-------------------
 *    22   | 22 + 21 + 21 points  | 2^22     | 8 (or 4)  | 192 GiB (96 GiB) | 49,152
 *    16   | 4 x 16               | 4 * 2^15 | 16        | 12.9 GB          | 65,536
 *     8   | 8 x 8                | 8 * 2^7  | 16        | 100.7 MB         | 131,072
 *     4   | 16 x 4               | 16 * 2^3 | 16        | 12.6 MB          | 262,144
 *
 * Class 22 also keeps a 403-MB latency comb (blocks of 8 points, 64 plane groups) for calls of at most 16 blobs.  Building
 * a table needs up to 13 GB of transient device scratch (XYZZ staging for the batch normalisation) on top of it.
 *
 * AUTOMATIC choice (window_bits = 0, what the Rust shim's Setup::load_json passes): the fastest class whose table is within
 * the caller's BUDGET and fits the HBM that is free when the context is created, with room left for the build scratch, the
 * call workspace and the caller's blobs:
 *     class 22, G = 8 (192 GiB)  if the budget allows it and >= 232 GiB are free
 *     class 22, G = 4 ( 96 GiB)  if the budget allows it and >= 136 GiB are free
 *     class 16        (12.9 GB)  if the budget allows it and >=  21 GiB are free
 *     class 8         (100 MB)   otherwise
 * The budget is table_budget_bytes; 0 means the DEFAULT budget of 100 GiB -- so an unconfigured context never takes more
 * than the 96-GiB table (2.4 % slower than the 192-GiB one), whatever the device has free -- unless flags carries
 * KZG_CFG_TABLE_MAX, which lifts the cap (bench.py opts in; 192 GiB on an idle MI355X).  An explicit window_bits /
 * plane_groups is honoured as given and fails if it cannot be built.  PRECEDENCE, one rule: a non-zero field of kzg_config
 * beats the environment (KATETH_AMD_WINDOW_BITS, KATETH_AMD_COMB_GROUPS, KATETH_AMD_TABLE_BUDGET_GIB -- an operator's cap on
 * the automatic choice of an unconfigured drop-in, e.g. 16 for the 12.9-GB table), which beats the automatic choice.
 * kzg_ctx_window_bits / kzg_ctx_plane_groups / kzg_ctx_table_bytes report what was built.
 */
#define KZG_CFG_TABLE_MAX 0x1   /* flags: the automatic choice may take the largest table the device has room for (192 GiB) */
#define KZG_CFG_BUILD_ASYNC 0x2 /* flags: kzg_ctx_create returns as soon as a small first-use table (class 8, 100 MB) stands and
                                 * builds the chosen table on a background thread, swapping it in between calls; results are
                                 * identical before and after the swap (kzg_ctx_ready / kzg_ctx_wait_ready) */
#define KZG_ALL_DEVICES 0xffffffffu /* ndev: every HIP device visible to the process */
typedef struct kzg_config {
  uint32_t struct_size; /* = sizeof(kzg_config) of the header the caller was compiled against (KZG_CONFIG_INIT sets it): the library
                         * rejects a size it does not know instead of reading fields the caller never wrote (the struct has grown
                         * between rounds and will again) */
  int32_t device;       /* HIP device ordinal this context lives on (ignored when ndev != 0) */
  int32_t window_bits;  /* table class: 22, 16..21 (-> 16), 8..15 (-> 8), 4..7 (-> 4); 0 = automatic (above) */
  int32_t flags;        /* KZG_CFG_* bits */
  int32_t plane_groups; /* G: 1, 2, 4, 8 or 16; 0 = automatic (class 22: 8 or 4 as above; the other classes: 16) */
  uint64_t table_budget_bytes; /* cap on the table the AUTOMATIC choice may build; 0 = default (100 GiB, or none with KZG_CFG_TABLE_MAX) */
  /* Multi-GPU (`kzg_ctx_create(g1, g2, devices[], ndev, &ctx)`): ndev != 0 makes the context a GROUP
   * with one member per listed ordinal (an ordinal may be listed more than once: two members on one card).  Every member
   * holds the full tables; the HOST-BUFFER entry points split a batch into contiguous ranges, one per member, run them on
   * one host thread per member and write the results straight into the caller's buffers (blobs are independent, src/kzg/
   * setup.rs:235-242: no collective); batch verification runs phase 1 / phase 2 per member with global indices, merges the
   * first-error records in the reference's order (src/kzg/setup.rs:259-271) and does ONE pairing check.  The *_dev entry
   * points, sessions, profiling and introspection act on member 0 (device pointers belong to one device);
   * kzg_ctx_member(ctx, k) lends member k's single-device context to callers that keep data resident on that GPU. */
  const int32_t* devices; /* ndev ordinals, or NULL with ndev = KZG_ALL_DEVICES */
  uint32_t ndev;          /* 0 = a single-device context on `device` */
  uint32_t reserved;      /* must be 0 */
} kzg_config;
#define KZG_CONFIG_INIT {(uint32_t)sizeof(kzg_config), 0, 0, 0, 0, 0, NULL, 0, 0} /* kzg_config cfg = KZG_CONFIG_INIT; then set fields */

/* "NAME=value" of the one environment variable of the HIP runtime the measured configurations depend on (see "Streams and
 * hardware queues" above); static storage.  The library itself never calls setenv. */
const char* kzg_recommended_env(void);

/* Thread-local text for the last negative return on this thread ("" if none). */
const char* kzg_last_error(void);
/* When that return was KZG_FAIL_SETUP_G1 / KZG_FAIL_SETUP_G2: the KZG_ERR_EC_* code of the rejected setup point, so that
 * the caller can rebuild LoadSetupError::Bls(bls::Error::ECGroup(..)) exactly (src/kzg/setup.rs:59-72); otherwise 0. */
int32_t kzg_last_error_code(void);

/*
 * Replaces Setup::<4096,65>::load_json after JSON/hex parsing
 * (src/kzg/setup.rs:52-81): decompresses and subgroup-checks the 4096 G1
 * Lagrange points and the 65 G2 monomial points (given in FILE order),
 * bit-reversal-permutes G1, derives the 4096 roots of unity (src/math.rs:16-29)
 * and builds the resident device tables.
 *   g1_lagrange : 4096 * 48 bytes      g2_monomial : 65 * 96 bytes
 */
int32_t kzg_ctx_create(const uint8_t* g1_lagrange, const uint8_t* g2_monomial, const kzg_config* cfg, kzg_ctx** out);
/* The same with the device list as arguments, the shape SURVEY.md section 8(b) wrote down: cfg's devices / ndev are replaced
 * by the arguments (cfg may be NULL); devices = NULL with ndev = KZG_ALL_DEVICES takes every visible device. */
int32_t kzg_ctx_create_multi(const uint8_t* g1_lagrange, const uint8_t* g2_monomial, const int32_t* devices, uint32_t ndev, const kzg_config* cfg,
                             kzg_ctx** out);
void kzg_ctx_destroy(kzg_ctx* ctx);
/* HIP devices visible to the process (what KZG_ALL_DEVICES expands to); negative = KZG_FAIL_* */
int32_t kzg_device_count(void);
/* members of a group context (1 for a single-device context), member k's device ordinal, and member k as a single-device
 * context (borrowed: owned and destroyed by `ctx`; member 0 of a single-device context is the context itself) */
uint32_t kzg_ctx_members(const kzg_ctx* ctx);
int32_t kzg_ctx_member_device(const kzg_ctx* ctx, uint32_t k);
const kzg_ctx* kzg_ctx_member(const kzg_ctx* ctx, uint32_t k);
/* KZG_CFG_BUILD_ASYNC: 1 once the chosen table is in use (always 1 without the flag), 0 while the first-use table serves;
 * kzg_ctx_wait_ready blocks until then and returns 0, or the KZG_FAIL_* code the background build ended with (the context
 * then stays on the first-use table).  On a group: all members. */
int32_t kzg_ctx_ready(const kzg_ctx* ctx);
int32_t kzg_ctx_wait_ready(const kzg_ctx* ctx);

/* introspection for benches / tests */
int32_t kzg_ctx_window_bits(const kzg_ctx* ctx);
const char* kzg_ctx_msm_kernel_name(const kzg_ctx* ctx); /* the dominant kernel bench.py names in `roofline` */
int32_t kzg_ctx_plane_groups(const kzg_ctx* ctx);
uint64_t kzg_ctx_table_bytes(const kzg_ctx* ctx);
/* bytes of commitment / proof workspace slot `slot` (0..2) as allocated so far: a caller that runs one call at a time only ever
 * grows slot 0; slots 1 and 2 are allocated when calls are found in flight side by side (tests, memory accounting) */
uint64_t kzg_ctx_workspace_bytes(const kzg_ctx* ctx, uint32_t slot);

/*
 * Replaces Setup::blob_to_commitment + Compress::compress for n blobs
 * (src/kzg/setup.rs:167-171, src/blob.rs:26-53, src/bls.rs:491-503).
 *   blobs  : n * 131072 bytes          out48  : n * 48 bytes
 *   status : n * int32 ; 0 or KZG_ERR_BLOB_INVALID_FIELD_ELEMENT per blob
 *            (a rejected blob's 48 output bytes are zeroed)
 * Host-pointer form copies in/out; the *_dev form takes HIP device pointers
 * resident on ctx's device and enqueues on `hip_stream` (a hipStream_t, NULL =
 * default stream) without synchronising.
 * `blob_len` lets the caller forward a wrong-length slice the way the Rust
 * shim would: blob_len != 131072 -> every status = KZG_ERR_BLOB_INVALID_LEN.
 */
int32_t kzg_blob_to_commitment_batch(const kzg_ctx* ctx, const uint8_t* blobs, uint64_t n, uint8_t* out48, int32_t* status);
int32_t kzg_blob_to_commitment_batch_dev(const kzg_ctx* ctx, const void* d_blobs, uint64_t n, void* d_out48, void* d_status, void* hip_stream);

/*
 * The same producers with the result as the reference returns it -- a POINT, `Commitment = Proof = P1`
 * (src/kzg/mod.rs:9-10; Setup::blob_to_commitment / blob_proof / proof, src/kzg/setup.rs:167,177,185) -- instead of its
 * 48-byte encoding: 96 bytes per item = blst_p1_affine, i.e. x || y, each 6 x uint64 little-endian limbs of the
 * 2^384-Montgomery residue (infinity = 96 zero bytes, as blst encodes it).  The Rust side rebuilds the `P1` with
 * blst_p1_from_affine -- no square root on the CPU -- and callers such as benches/kzg.rs:24-32 can keep calling
 * `.compress()` on it.  Rejected items get 96 zero bytes and their status code.
 */
int32_t kzg_blob_to_commitment_batch_affine(const kzg_ctx* ctx, const uint8_t* blobs, uint64_t n, uint8_t* out_affine96, int32_t* status);
int32_t kzg_compute_blob_proof_batch_affine(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* commitments48, uint64_t n, uint8_t* out_affine96, int32_t* status);
int32_t kzg_compute_proof_batch_affine(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* z32, uint64_t n, uint8_t* out_proof_affine96, uint8_t* out_y32, int32_t* status);

/*
 * Replaces P1::decompress -- the crate's public `Decompress` trait on G1 points, i.e. on `Commitment` / `Proof`
 * (src/lib.rs:5, src/bls.rs:505-531: ZCash decoding, on-curve check, SUBGROUP check) -- for n compressed points:
 *   in48         : n * 48 bytes
 *   out_affine96 : n * 96 bytes, blst_p1_affine images as above (infinity = 96 zero bytes, status 0)
 *   status       : per point 0 or KZG_ERR_EC_INVALID_ENCODING / KZG_ERR_EC_NOT_ON_CURVE / KZG_ERR_EC_NOT_IN_GROUP
 *                  (a rejected point's 96 output bytes are zeroed)
 * The same decoder the verification entry points run on their commitments and proofs.
 */
int32_t kzg_g1_decompress_batch(const kzg_ctx* ctx, const uint8_t* in48, uint64_t n, uint8_t* out_affine96, int32_t* status);

/*
 * Replaces Polynomial::evaluate (src/kzg/poly.rs:10-33; with Blob::from_slice, src/blob.rs:26-37) for n (blob, z) pairs --
 * the evaluation step of the verification path on its own (there z is a hash output; an evaluation point on the domain,
 * poly.rs:14-18, reaches that kernel only through this call).  Any n: the blobs stream through the staging arena in chunks.
 *   blobs   : n * 131072 bytes          z32 : n * 32 bytes, big-endian canonical
 *   out_y32 : n * 32 bytes, big-endian (zero bytes for a rejected item)
 *   status  : per item 0, KZG_ERR_BLOB_INVALID_FIELD_ELEMENT, or KZG_ERR_FF_NOT_IN_FIELD for z
 */
int32_t kzg_evaluate_blobs(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* z32, uint64_t n, uint8_t* out_y32, int32_t* status);

/*
 * Replaces Setup::blob_proof + compress for n (blob, commitment) pairs
 * (src/kzg/setup.rs:177-183, src/blob.rs:55-97, src/kzg/poly.rs:10-71).
 *   status : per item 0, KZG_ERR_BLOB_*, or KZG_ERR_EC_* for the commitment
 */
int32_t kzg_compute_blob_proof_batch(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* commitments48, uint64_t n, uint8_t* out48, int32_t* status);
int32_t kzg_compute_blob_proof_batch_dev(const kzg_ctx* ctx, const void* d_blobs, const void* d_commitments48, uint64_t n, void* d_out48, void* d_status, void* hip_stream);

/*
 * Replaces Setup::proof for n (blob, z) pairs (src/kzg/setup.rs:185-194):
 *   z32 : n * 32 bytes big-endian ; out_proof48 : n * 48 ; out_y32 : n * 32
 *   status : per item 0, KZG_ERR_BLOB_*, KZG_ERR_FF_NOT_IN_FIELD for z
 */
int32_t kzg_compute_proof_batch(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* z32, uint64_t n, uint8_t* out_proof48, uint8_t* out_y32, int32_t* status);

/*
 * Replaces Setup::verify_blob_proof_batch (src/kzg/setup.rs:247-275).
 * Returns KZG_OK with *ok = 0/1, or -- like the reference's first-error-wins
 * collect (src/kzg/setup.rs:259-271: all blobs, then all commitments, then all
 * proofs) -- the KZG_ERR_* of the first rejected input, *ok = 0.
 * n == 0 -> *ok = 1.  The length-equality assert of the reference
 * (src/kzg/setup.rs:256-257) stays in the caller: the ABI takes a single n.
 */
int32_t kzg_verify_blob_proof_batch(const kzg_ctx* ctx, const uint8_t* blobs, const uint8_t* commitments48, const uint8_t* proofs48, uint64_t n, int32_t* ok);
int32_t kzg_verify_blob_proof_batch_dev(const kzg_ctx* ctx, const void* d_blobs, const void* d_commitments48, const void* d_proofs48, uint64_t n, int32_t* ok, void* hip_stream);

/* Replaces Setup::verify_blob_proof (src/kzg/setup.rs:208-221): one item. */
int32_t kzg_verify_blob_proof(const kzg_ctx* ctx, const uint8_t* blob, const uint8_t* commitment48, const uint8_t* proof48, int32_t* ok);

/* Replaces Setup::verify_proof (src/kzg/setup.rs:96-113). */
int32_t kzg_verify_proof(const kzg_ctx* ctx, const uint8_t* proof48, const uint8_t* commitment48, const uint8_t* z32, const uint8_t* y32, int32_t* ok);

/*
 * DEVICE-RESIDENT sharded calls on a GROUP context (kzg_config.devices / ndev): member k's share of the batch is resident on
 * member k's GPU -- what a node keeps when blobs arrive over the network or are produced on the devices, and the only way a
 * group verifies faster than PCIe delivers (host-buffer verification tops out near 0.37 M blobs/s per GPU, device-resident
 * verification runs at 4.4 M).  Every array argument has kzg_ctx_members(ctx) entries, indexed by member; the GLOBAL order of
 * the batch is member order (member 0's n_local[0] items first), which is what the powers r^i of the batch check and the
 * first-error-wins order (src/kzg/setup.rs:259-271: lowest global index of the first kind with an error) are taken over.
 * n_local[k] may be 0.  hip_streams may be NULL (every member's default stream) or one hipStream_t per member, created on that
 * member's device.
 *   commit / proof : enqueue on every member and return WITHOUT synchronising, like the *_dev calls
 *                    (src/kzg/setup.rs:167-171, 177-183 per item); results and per-item statuses stay on each member's device.
 *   verify         : synchronous like kzg_verify_blob_proof_batch_dev: per member hash, evaluation, decoding and transcript,
 *                    the members' roots seed ONE challenge, per member the two partial lincombs (bucket kernels start the
 *                    moment that member's decoder ends, as in the single-device call), one pairing check
 *                    (src/kzg/setup.rs:223-275).  Same boolean and same first-error code as the single-device call over the
 *                    concatenated batch.
 * On a single-device context they are the *_dev calls with one-element arrays.
 */
int32_t kzg_blob_to_commitment_batch_group_dev(const kzg_ctx* ctx, const void* const* d_blobs, const uint64_t* n_local, void* const* d_out48,
                                               void* const* d_status, void* const* hip_streams);
int32_t kzg_compute_blob_proof_batch_group_dev(const kzg_ctx* ctx, const void* const* d_blobs, const void* const* d_commitments48, const uint64_t* n_local,
                                               void* const* d_out48, void* const* d_status, void* const* hip_streams);
int32_t kzg_verify_blob_proof_batch_group_dev(const kzg_ctx* ctx, const void* const* d_blobs, const void* const* d_commitments48,
                                              const void* const* d_proofs48, const uint64_t* n_local, int32_t* ok, void* const* hip_streams);

/*
 * Multi-GPU batch verification.  Blobs are sharded by
 * contiguous global index ranges; each rank runs
 *   phase 1 : per-item work on its n_local items (validation, challenge z_i,
 *             evaluation y_i) and a 32-byte transcript root over them;
 *             err6 = {blob_idx, blob_code, commitment_idx, commitment_code,
 *             proof_idx, proof_code} with LOCAL index of the first rejected
 *             input of each kind (-1 = none), so the caller can rebuild the
 *             reference's first-error-wins order (src/kzg/setup.rs:259-271)
 *             across ranks;
 *   (the caller gathers the `world` roots -- RCCL all-gather of 32 B per rank)
 *   phase 2 : the rank's share of the two random linear combinations, with the
 *             challenge seeded by ALL roots; 2 x 96 bytes (affine big-endian
 *             x||y, all-zero = infinity);
 *   finish  : on one rank, sums the gathered partials (192 B per rank) and runs
 *             the single two-pairing check.
 * A session owns its device buffers and stream and may not be used concurrently; kzg_verify_session_destroy hands it
 * back to the context's pool (steady-state verification allocates nothing).
 */
typedef struct kzg_verify_session kzg_verify_session;
int32_t kzg_verify_phase1_dev(const kzg_ctx* ctx, const void* d_blobs, const void* d_commitments48, const void* d_proofs48, uint64_t n_local,
                              uint8_t* out_root32, int32_t* err6, kzg_verify_session** session, void* hip_stream);
int32_t kzg_verify_phase2_dev(kzg_verify_session* session, const uint8_t* roots32, uint64_t world, uint64_t first_index, uint64_t n_total,
                              uint8_t* out192);
void kzg_verify_session_destroy(kzg_verify_session* session);
/* introspection (tests, debugging): the Fiat-Shamir challenges z_i (Blob::challenge, src/blob.rs:78-97) and evaluations
 * y_i (Polynomial::evaluate, src/kzg/poly.rs:10-33) phase 1 computed for items [first, first+count), 32 B big-endian each */
int32_t kzg_verify_session_zy(kzg_verify_session* session, uint64_t first, uint64_t count, uint8_t* out_z32, uint8_t* out_y32);
int32_t kzg_verify_batch_finish(const kzg_ctx* ctx, const uint8_t* partials192, uint64_t world, int32_t* ok);

/*
 * Synthetic-input generator used by bench.py and the parity tests
 * (counterpart of Blob::random, src/blob.rs:66-76, but seeded):
 *   element(b, i) = SHA-256(seed_le64 || b_le64 || i_le32) mod r, 32 B big-endian
 * Fills n blobs with indices first_index .. first_index+n-1 into d_blobs.
 */
int32_t kzg_synth_blobs_dev(const kzg_ctx* ctx, uint64_t seed, uint64_t first_index, uint64_t n, void* d_blobs, void* hip_stream);

/*
 * Device micro-benchmarks (measurement support for bench.py's roofline object):
 * runs `iters` dependent Fp Montgomery multiplications per lane on `lanes`
 * lanes -- with the multiply of the fixed-base MSM kernel the context uses
 * (radix-2^28 limbs by default) -- and returns the elapsed milliseconds
 * measured with HIP events.
 */
int32_t kzg_microbench_fp_mul(const kzg_ctx* ctx, uint64_t lanes, uint64_t iters, float* ms);
/*
 * The VALU issue interval the roofline is priced with: SIMD cycles per wave-instruction of v_mad_u64_u32 (the
 * instruction 76-80 % of the MSM / evaluation / decoding streams consist of) with `waves_per_simd` (1..4) co-resident
 * waves on every SIMD of the chip, 8 independent chains per wave, `iters` x 32 instructions per lane (median over the
 * waves, by the shader-clock counter), and the shader clock in GHz the chip sustains under that load (shader-clock
 * ticks per 100-MHz real-time tick).  floor of a kernel = SQ_INSTS_VALU / SIMDs x cycles_per_inst / clock.
 */
int32_t kzg_microbench_valu_issue(const kzg_ctx* ctx, uint32_t waves_per_simd, uint32_t iters, double* cycles_per_inst, double* clock_ghz);
/*
 * The shader clock under a REAL workload: kzg_clock_probe_launch enqueues eight sleeping single-wave workgroups (one per
 * XCD) on a stream of the context's own; for `duration_us` they compare the shader-clock counter with the 100-MHz real-time
 * counter while the caller's kernels run beside them (a probe wave needs a handful of registers and issues one instruction
 * every 2 us).  kzg_clock_probe_read waits for them and returns the mean / lowest / highest XCD clock in GHz.
 */
int32_t kzg_clock_probe_launch(const kzg_ctx* ctx, uint32_t duration_us);
int32_t kzg_clock_probe_read(const kzg_ctx* ctx, double* ghz_mean, double* ghz_min, double* ghz_max);

/*
 * On-device self-test of the hand-scheduled multiply: every lane multiplies
 * `iters` pseudo-random operand pairs with the inline-asm v_mad_u64_u32 chains
 * and with a plain-C multiply the compiler schedules (hazard wait states and
 * all); *mismatches counts disagreements (Fp and Fr).  The same operands also go
 * through the carry-free radix-2^28 (Fp) and radix-2^29 (Fr) multipliers of the
 * hot loops (product, squaring, two products with one reduction) and must agree
 * with the 32-bit-limb result.  Expected: 0.
 */
int32_t kzg_selftest_field_mul(const kzg_ctx* ctx, uint64_t lanes, uint64_t iters, uint64_t* mismatches);
/* Test hook for the guard described at KZG_FAIL_HOST: throws std::bad_alloc (kind 0), std::runtime_error (1) or an int (2) inside
 * an entry point; expected return KZG_FAIL_HOST with the exception's text in kzg_last_error().  Touches no GPU. */
int32_t kzg_selftest_exception_guard(int32_t kind);

/*
 * Kernel timing for bench.py's `roofline` object: between begin and end every
 * launch of the timed kernel classes (below) is bracketed by HIP events on the
 * stream it is launched on.  end() synchronises those events and returns the
 * summed milliseconds and the number of launches of the fixed-base MSM kernel
 * (k_msm_comb30).  A profiling interval must not overlap calls still being
 * enqueued from other threads (their unfinished event pairs are skipped).
 */
int32_t kzg_profile_begin(const kzg_ctx* ctx);
int32_t kzg_profile_end(const kzg_ctx* ctx, double* msm_ms_total, uint64_t* msm_launches);
/*
 * The same for every timed kernel class: ms[k] / launches[k] for k < KZG_PROF_KINDS (kzg_profile_kind_name(k) names
 * the class: the fixed-base MSM, the SHA-256 challenge, the barycentric evaluation, point decoding, the quotient
 * kernel, the variable-base MSMs of batch verification, lane-sum trees + encoding).  Ends the profiling interval.
 */
#define KZG_PROF_KINDS 8
int32_t kzg_profile_end_kinds(const kzg_ctx* ctx, double* ms, uint64_t* launches);
const char* kzg_profile_kind_name(int32_t kind);
/* mixed additions the fixed-base MSM performs per blob: 256 bit planes x (blocks per 64 points) x 64 (table above) */
uint64_t kzg_ctx_adds_per_blob(const kzg_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* KATETH_AMD_H */
