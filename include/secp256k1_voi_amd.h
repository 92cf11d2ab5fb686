/*
 * secp256k1_voi_amd.h — C-ABI of the MI355X (gfx950) batch engine for the verify /
 * scalar-multiplication path of Yawning/secp256k1-voi.
 *
 * The reference is a pure-Go library with no FFI of its own; its
 * boundary is the exported Go API.  This header is what a cgo shim that keeps the
 * reference's Point / Scalar / secec.Verify surface binds to (INTEGRATION.md shows the
 * shim).  Every entry point names the reference symbol it serves (file:line relative to
 * the reference tree).
 *
 * Conventions (identical to the reference's canonical encodings):
 *   - scalars and field elements: 32-byte big-endian (Scalar.Bytes scalar.go:148,
 *     Element.Bytes field.go:151);
 *   - points, where a full point crosses the boundary: fixed 65-byte records, either
 *     0x04‖X‖Y (Point.UncompressedBytes, point_s11n.go:66) or 0x00 followed by 64 zero
 *     bytes for the identity (the reference's 1-byte identity encoding, padded);
 *   - public keys of the verify path: 64-byte X‖Y (the 65-byte SEC1 form minus the 0x04
 *     prefix, i.e. PublicKey.Bytes()[1:], secec.go:80-85);
 *   - per-item results are data (0/1 bytes), never errors — like Verify returning false
 *     (ecdsa.go:186-188).  The int return value reports infrastructure errors only.
 *   - all buffers are caller-owned; nothing is retained after a call returns.
 *   - a context is bound to one GPU and owns the device workspaces its calls share.  HOST side:
 *     calls on one context must not run concurrently (one context per goroutine/thread, like
 *     distinct receivers in the reference).  DEVICE side: the *_device entry points enqueue on
 *     the caller's stream and return without synchronising; the context chains consecutive
 *     calls with an event, so a call enqueued on another stream (or a host-buffer entry point,
 *     which uses the context's own streams) starts only after the previous call's kernels have
 *     finished — callers may switch streams freely, results are never torn.
 *
 * There is no CPU fallback: every function fails with S2K_ERR_NO_DEVICE / S2K_ERR_HIP
 * when the GPU is not usable.
 */
#ifndef SECP256K1_VOI_AMD_H
#define SECP256K1_VOI_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct s2k_ctx s2k_ctx;

enum {
  S2K_OK = 0,
  S2K_ERR_NO_DEVICE = -1, /* no usable gfx950 device */
  S2K_ERR_HIP = -2,       /* a HIP runtime call failed; see s2k_last_error */
  S2K_ERR_ARG = -3,       /* null pointer / bad length / length mismatch (the reference panics: point_mul_multi.go:27) */
  S2K_ERR_NOMEM = -4,
  S2K_PENDING = 1         /* s2k_poll: the ticket is still in flight (not an error) */
};

/* flags for the ECDSA entry points */
#define S2K_ECDSA_REJECT_MALLEABLE 1u /* ECDSAOptions.RejectMalleable: reject s > n/2 (ecdsa.go:212) */
/* Diagnostics: send every signature through the complete-formula kernel that normally only
 * re-does the lanes the fast Jacobian kernel cannot decide.  Same results, ~3x slower. */
#define S2K_ECDSA_FORCE_COMPLETE 0x80000000u
/* Diagnostics: the fast ladder runs, then EVERY signature is queued for the complete-formula worklist kernel
 * (9x29 field, what decides lanes whose incomplete formulas met an exceptional case inside the ladder).  Same results,
 * ~3.4x a normal step: the price of a batch in which every lane is such a case.  The exceptional cases a key owner can
 * force for given digests - u1 G = -u2 Q (r = -e/d: decided inside the ladder kernel) and u1 G = u2 Q (r = e/d: a short
 * form of the worklist kernel, 2 u1 G) - arise in the final addition and no longer cost this; u1 = 0 needs a digest that
 * is 0 mod n.  One crafted value of u2 = r/s per ladder form used to make the ladder's last table addition add a point to
 * itself; the odd GLV split avoids it now (bench key worst_case_ladder_collision).  No input is known
 * that fills the worklist any more; the flag keeps the kernel measurable and tested. */
#define S2K_ECDSA_FORCE_WORKLIST 0x40000000u

/* bitcoin.VerifyASN1 (secec/bitcoin/ecdsa_shitcoin.go:29): BIP-0066 shape check, sighash byte
 * stripped, low-s required, 32-byte digest.  Only for s2k_ecdsa_verify_encoded_batch. */
#define S2K_ECDSA_BIP0066 2u

/* SignatureEncoding (ecdsa.go:39-53) */
#define S2K_ENCODING_ASN1 0
#define S2K_ENCODING_COMPACT 1
#define S2K_ENCODING_COMPACT_RECOVERABLE 2 /* [R | S | V], 65 bytes: ParseCompactRecoverableSignature (s11n.go:156-168) */

#define S2K_POINT_RECORD 65
#define S2K_SCALAR_SIZE 32 /* ScalarSize, scalar.go:17 */
#define S2K_COORD_SIZE 32  /* CoordSize, point_s11n.go:27-44 */

/* ---- context --------------------------------------------------------------------- */
/* Selects the device and makes the resident generator tables available (the device analogue of the package-init unpack of
 * generatorHugeAffineTable, point_mul_table.go:75-100).  The tables of a device are shared by the contexts of a process.
 * s2k_ctx_create returns as soon as a narrow table (20-bit windows, 0.8 GiB) is built - under 0.1 s - and a background
 * thread builds the wide one (26-bit windows, 40 GiB, 3 s; 24 or 22 bits when the device has less than twice that free or
 * s2k_set_generator_table_budget says so; none when even those do not fit): calls made meanwhile run on the narrow table
 * (about 2 % slower), the first CALL that starts after the build uses the wide one (a call never changes tables between its
 * launches).  Verdicts do not depend on the width.  A context's child contexts (submit / wait) use its width.
 * s2k_ctx_create_ex: gt_bits 0 = as above; 16 .. 26 = tables of exactly that width for this context, built before the call
 * returns (S2K_ERR_NOMEM when the device cannot hold them); flags S2K_CTX_WAIT_TABLES = automatic width, but return only
 * when the background build has ended.  s2k_ctx_gt_info: info[0] window bits in use now, [1] bits being aimed for (0: none),
 * [2] 1 while the build runs, [3] bytes of tables the device holds for this process; s2k_ctx_gt_note: why (free memory,
 * budget, a failed allocation); s2k_ctx_gt_wait: block until the build has ended, returns the bits in use then.
 * The builder thread is never left behind: the s2k_ctx_destroy that removes the last context of a device cancels it (it
 * looks at the flag before it allocates, between the windows of a table and before it publishes) and joins it - 0.01-0.13 s,
 * or the rest of the 40 GiB allocation when the destroy meets it (1.6 s; profiles/r06_ctx_lifecycle.txt) - and a process that
 * exits with contexts alive does the same from an atexit handler, before the HIP runtime is torn down. */
#define S2K_CTX_WAIT_TABLES 1u
int s2k_ctx_create(int device_index, s2k_ctx **out);
int s2k_ctx_create_ex(int device_index, int gt_bits, uint32_t flags, s2k_ctx **out);
void s2k_set_generator_table_budget(size_t bytes_per_device);
/* Caps for the other two table kinds of this process (0 = none): the bytes s2k_keyset_create_ex may count as free when it chooses
 * (S2K_KEYSET_AUTO) or checks a joint-table layout, and the largest per-key table buffer a verification call may allocate (a
 * larger request is treated as a failed allocation: the call degrades as it does under real memory pressure). */
void s2k_set_table_memory_budgets(size_t keyset_free_bytes, size_t key_table_bytes);
int s2k_ctx_gt_info(s2k_ctx *ctx, uint64_t info[4]);
const char *s2k_ctx_gt_note(s2k_ctx *ctx);
int s2k_ctx_gt_wait(s2k_ctx *ctx);
/* Test hook: the next s2k_ecdsa_verify_batch_device call of this (automatic) context makes the table of `bits` the one the
 * device's automatic contexts use BETWEEN its ladder launch and its worklist launch - the worst moment for the background
 * build to publish.  A call uses one table for all its launches, so its verdicts must not change.  One shot; 0 disarms. */
int s2k_debug_gt_swap_in_call(s2k_ctx *ctx, int bits);
void s2k_ctx_destroy(s2k_ctx *ctx);
const char *s2k_last_error(const s2k_ctx *ctx);
const char *s2k_version(void);
/* Compile-time configuration of this library ("GT_BITS=26 PREP_M=6 ... flags=<S2K_EXTRA_FLAGS>"),
 * so that a measurement can name the variant it ran (bench.py prints it). */
const char *s2k_build_config(void);
/* Measurement hooks (no reference counterpart; bench.py's live roofline).  While enabled, every
 * s2k_ecdsa_verify_batch_device call records HIP events on its stream around its three kernels
 * (scalar preparation, ladder, complete-formula worklist) and the ladder kernel stamps the shader
 * cycle counter against the constant-rate wall clock.  s2k_ctx_profile_read synchronises the
 * device and returns the summed durations in ms_sum3[0..2], optionally each call's ladder time
 * in ms_fast_each[0..cap), the number of calls since the last read, and the effective shader
 * clock (MHz) of the last ladder launch: shader_mhz[0] from its first wave, shader_mhz[1] from a wave
 * of its final round (the clock sags under power during a launch). */
int s2k_ctx_profile(s2k_ctx *ctx, int enable);
int s2k_ctx_profile_read(s2k_ctx *ctx, double ms_sum3[3], double *ms_fast_each, size_t cap, size_t *calls,
                         double shader_mhz[2]);
/* The same with the stages apart.  Grouping off: ms_sum5[0] scalar preparation, [2] ladder, [4]
 * complete-formula worklist.  Grouping on: [0] grouping by key, [1] per-key tables with the scalar
 * preparation running beside them, [2] ladder over the per-key tables, [3] general ladder over the
 * remaining signatures, [4] worklist.  ms_fast_each is stage [2]. */
int s2k_ctx_profile_read_stages(s2k_ctx *ctx, double ms_sum5[5], double *ms_fast_each, size_t cap, size_t *calls,
                                double shader_mhz[2]);

/* The same for the multi-scalar path (s2k_multi_scalar_mult_device, s2k_schnorr_batch_verify_rlc_device and the
 * bisection's sub-range runs): while enabled every call records HIP events on its stream between its stages;
 * s2k_ctx_profile_read_msm synchronises the device and returns the summed durations: ms_sum5[0] front end (parsing of
 * the terms; for BIP-340 the per-signature preparation with the key grouping, key lifts and key terms), [1] sort of the
 * (window, digit) keys, [2] bucket pass (k_msm_accumulate: the dominant kernel), [3] stitching of the buckets, bucket
 * reduction and tree, [4] Horner tail and affine result; and the number of calls since the last read. */
int s2k_ctx_profile_msm(s2k_ctx *ctx, int enable);
int s2k_ctx_profile_read_msm(s2k_ctx *ctx, double ms_sum5[5], size_t *calls);

/* ---- hot path: batch ECDSA verification ------------------------------------------- */
/* For each i < n: secec.PublicKey.VerifyRaw(digest, r, s) (ecdsa.go:234 -> verify :392)
 * after ParseCompactSignature's range checks (s11n.go:129-144) and NewPublicKey's point
 * validation (secec.go:188-216, point_s11n.go:178-209), all done on the device:
 *   valid[i] = 1 iff r,s in [1,n), (flags&REJECT_MALLEABLE => s <= n/2), (X,Y) canonical
 *              and on the curve, R = u1*G + u2*Q != identity and x(R) mod n == r,
 *   with e = digest32 reduced mod n (hashToScalar, ecdsa.go:477-486; the caller passes
 *   the leftmost 32 bytes of its digest).
 * Host-pointer form: copies in, runs, copies out. */
int s2k_ecdsa_verify_batch(s2k_ctx *ctx, size_t n, const uint8_t *pub_xy /* n*64 */,
                           const uint8_t *digest32 /* n*32 */, const uint8_t *r /* n*32 */,
                           const uint8_t *s /* n*32 */, uint32_t flags, uint8_t *valid /* n */);
/* Device-pointer form: all pointers are device memory (16-byte aligned); enqueues on
 * `hip_stream` (a hipStream_t, NULL = default stream) and returns without synchronising. */
int s2k_ecdsa_verify_batch_device(s2k_ctx *ctx, size_t n, const void *d_pub_xy, const void *d_digest32,
                                  const void *d_r, const void *d_s, uint32_t flags, void *d_valid,
                                  void *hip_stream);
/* ---- signatures that share public keys ---------------------------------------------------- */
/* The reference keeps nothing per key (secec.PublicKey holds the point and its encoding, secec.go:150;
 * every Verify starts DoubleScalarMultBasepointVartime from Q, point_mul_glv.go:307).  Here a batch
 * is first grouped by public key on the device (exact: a hash table keyed by the 64 key bytes, full
 * comparison on every hit); a key with at least `min_group` signatures gets one precomputed table,
 * built inside the call, after which each of its signatures costs 12 point doublings instead of 128;
 * the other signatures take the general kernel.  Verdicts are the same either way.
 *   mode        S2K_KEYS_OFF: every signature through the general kernel (the round-1/2 path)
 *               S2K_KEYS_AUTO: as described, in every call
 *               S2K_KEYS_ALWAYS: tables even for keys with a single signature (tests)
 *               S2K_KEYS_ADAPTIVE (default): S2K_KEYS_AUTO that stops looking when there is nothing to find.  A batch
 *               whose keys never repeat pays 1-2 % for the grouping that finds nothing; the context therefore notes what
 *               each grouped call of at least 2^16 signatures found (one word written by the device into page-locked host
 *               memory, read by later calls: no synchronisation), and after two consecutive observed calls in which no
 *               key reached the threshold it verifies the next 15 such batches as S2K_KEYS_OFF does, looks again in
 *               one, and so on; the first observed call that finds a group ends the skipping.  A workload that turns
 *               from distinct keys to repeated ones is therefore verified at the general ladder's rate for up to 15
 *               calls (plus the calls already enqueued) before its tables come back.  What has been learned belongs to
 *               the context and survives mode changes; verdicts do not depend on it (s2k_ctx_key_grouping_adaptive)
 *   min_group   0 = default (4): measured break-even between 3 and 4 signatures per key
 *   hash_bits   0 = default (slots >= 2n); smaller values force probe chains (tests)
 *   max_tables  0 = default (2^22 tables of 9 KiB; the buffer is sized by the batch: n / min_group tables).  The threshold is raised until n / threshold tables
 *               fit: a batch of n signatures builds tables for keys with at least
 *               max(min_group, ceil(n / max_tables)) signatures.  A device that has no memory for the buffer does not
 *               fail the verification: the cap is halved until the buffer fits, and below 1024 tables the batch is
 *               verified without tables (same verdicts) */
#define S2K_KEYS_OFF 0
#define S2K_KEYS_AUTO 1
#define S2K_KEYS_ALWAYS 2
#define S2K_KEYS_ADAPTIVE 3
int s2k_ctx_set_key_grouping(s2k_ctx *ctx, int mode, uint32_t min_group, uint32_t hash_bits, uint32_t max_tables);
/* After the last s2k_ecdsa_verify_batch_device call has finished (synchronises the device):
 * stats[0] signatures verified from per-key tables, [1] tables built, [2] signatures through the
 * general kernel, [3] signatures re-done by the complete-formula kernel.  All zero after a call that took the ladders of
 * small or mid-size batches (s2k_ctx_set_small_batch_max / s2k_ctx_set_mid_batch_max): they neither group nor have a worklist. */
int s2k_ctx_key_grouping_stats(s2k_ctx *ctx, uint32_t stats[4]);
/* State of S2K_KEYS_ADAPTIVE, without synchronising: out[0] consecutive observed calls that found no group, [1] calls
 * still to be verified without looking, [2] calls verified without looking so far, [3] calls that looked again after a
 * run of skipped ones, [4] grouped calls whose result has been observed.  reset != 0: forget what was learned. */
int s2k_ctx_key_grouping_adaptive(s2k_ctx *ctx, uint32_t out[5], int reset);

/* ---- batch public-key recovery ------------------------------------------------------------ */
/* For each i < n: secec.RecoverPublicKey(digest, r, s, recovery_id) (ecdsa.go:244-282):
 * R = RecoverPoint(r, id) (point_s11n.go:245-282), Q = (-e/r) G + (s/r) R.  ok[i] = 1 and
 * pub65[i] = 0x04 || X || Y on success; ok[i] = 0 and a zero record when the reference returns
 * an error (r or s not in [1, n), id > 3, x = r + n not below p, x not on the curve, Q = infinity).
 * Like the reference, s > n/2 is accepted here.  flags: S2K_ECDSA_FORCE_COMPLETE only. */
int s2k_ecdsa_recover_batch(s2k_ctx *ctx, size_t n, const uint8_t *digest32 /* n*32 */, const uint8_t *r /* n*32 */,
                            const uint8_t *s /* n*32 */, const uint8_t *recovery_id /* n */, uint32_t flags,
                            uint8_t *pub65 /* n*65 */, uint8_t *ok /* n */);
int s2k_ecdsa_recover_batch_device(s2k_ctx *ctx, size_t n, const void *d_digest32, const void *d_r, const void *d_s,
                                   const void *d_recovery_id, uint32_t flags, void *d_pub65, void *d_ok,
                                   void *hip_stream);

/* ---- key sets: per-key precomputation kept across calls ------------------------------- */
/* The reference caches per-key state in secec.PublicKey (secec/secec.go:80-85: the decoded point and its encodings,
 * built once by NewPublicKey, :188-216) and every Verify starts from it.  The device analogue: a KEY SET holds, for a
 * fixed list of n_keys public keys (X || Y, 64 bytes each), the per-key tables the verification ladder otherwise
 * builds inside every call for keys that repeat in the batch (9 KiB per key; a key that is not a valid public key gets
 * a table entry marked invalid: every signature under it verifies false, as NewPublicKey would have refused the key).
 * s2k_ecdsa_verify_batch_keyset[_device]: valid[i] = PublicKey(keys[key_index[i]]).VerifyRaw(digest_i, r_i, s_i)
 * (ecdsa.go:234) with the same rules as s2k_ecdsa_verify_batch; key_index[i] >= n_keys names no key: valid[i] = 0.
 * Same verdicts as s2k_ecdsa_verify_batch on the expanded key array, bit for bit; what is saved is the grouping and the
 * table build of every call (about a third of a 2^20-signature step at 16 signatures per key).  A key set belongs to
 * the context that made it (a context destroyed and another created at the same address is NOT the owner: the set
 * remembers the context's generation and device); it may be used by any number of calls and must be destroyed before the
 * context.  Like every object of a context it is not locked: s2k_keyset_destroy / s2k_keyset_valid_keys must not run
 * while a call that uses the set is in progress on another thread. */
typedef struct s2k_keyset s2k_keyset;
int s2k_keyset_create(s2k_ctx *ctx, size_t n_keys, const uint8_t *pub_xy /* n_keys*64, host */, s2k_keyset **out);
/* The same with the table layout named.  A key set is built once, so its tables can be large:
 *   S2K_KEYSET_CHUNKS  one 8-entry chunk per 4-bit digit of a half scalar (36 KiB per key): a signature's ladder is 64 table
 *                      additions and no doubling;
 *   S2K_KEYSET_JOINT   on top of that, per digit position the sums E_a + s phi(E_b) of the chunk's entries and their images
 *                      under the endomorphism (320 KiB per key more): the two half scalars' digits at a position are ONE table
 *                      addition, 32 per signature;
 *   S2K_KEYSET_JOINT5  the same on 5-bit digits: 26 digit positions, 512 sums per position - 26 table additions per signature,
 *                      0.81 MiB per key on top of the chunk tables (56 GB for 2^16 keys);
 *   S2K_KEYSET_JOINT6  on 6-bit digits: 22 positions, 2048 sums each - 22 additions, 2.75 MiB per key (for sets of up to
 *                      2^15 keys or so on a 288 GB device);
 *   S2K_KEYSET_AUTO    (s2k_keyset_create) the widest of JOINT5 and JOINT that takes no more than half of the device memory
 *                      free at the time, else chunks.
 * Verdicts are identical in every layout.  s2k_keyset_layout tells which one a set has. */
#define S2K_KEYSET_AUTO 0
#define S2K_KEYSET_CHUNKS 1
#define S2K_KEYSET_JOINT 2
#define S2K_KEYSET_JOINT5 3
#define S2K_KEYSET_JOINT6 4
int s2k_keyset_create_ex(s2k_ctx *ctx, size_t n_keys, const uint8_t *pub_xy, int layout, s2k_keyset **out);
int s2k_keyset_layout(const s2k_keyset *ks);
void s2k_keyset_destroy(s2k_keyset *ks);
size_t s2k_keyset_size(const s2k_keyset *ks);
size_t s2k_keyset_device_bytes(const s2k_keyset *ks);
/* valid[k] = 1 iff key k is a valid public key (canonical coordinates, on the curve): NewPublicKey's verdict per key. */
int s2k_keyset_valid_keys(s2k_keyset *ks, uint8_t *valid /* n_keys, host */);
int s2k_ecdsa_verify_batch_keyset(s2k_ctx *ctx, const s2k_keyset *ks, size_t n, const uint32_t *key_index /* n */,
                                  const uint8_t *digest32, const uint8_t *r, const uint8_t *s, uint32_t flags, uint8_t *valid);
int s2k_ecdsa_verify_batch_keyset_device(s2k_ctx *ctx, const s2k_keyset *ks, size_t n, const void *d_key_index /* n uint32 */,
                                         const void *d_digest32, const void *d_r, const void *d_s, uint32_t flags,
                                         void *d_valid, void *hip_stream);

/* Page-locked host buffers for the host-pointer entry points (no reference counterpart: the reference never leaves the
 * host; this is the price of the boundary, cf. secec.PublicKey.Verify, ecdsa.go:171-228, which a cgo shim batches into
 * s2k_ecdsa_verify_batch).  From pageable memory every copy is staged by the runtime and 2^20 verifications cost
 * 8.4-9.5 ms against 5.2-5.7 resident; from pinned memory the copies are asynchronous and the batch is processed in one
 * grouped call whose table phase overlaps the transfer of the digests and signatures.  s2k_host_alloc / s2k_host_free:
 * hipHostMalloc / hipHostFree (NULL on failure).  s2k_host_register / s2k_host_unregister: pin memory the caller
 * already owns; unregister before the memory is freed or moved.  The buffer must start on a page boundary and cover
 * whole pages (S2K_ERR_ARG otherwise): the runtime pins and unmaps pages, and a block of the C or Go heap shares its
 * first and last page with its neighbours - on ROCm 7.2 a later pageable copy from such a re-used page takes the
 * process down with a GPU memory access fault.  Memory from mmap / aligned_alloc(page, k * page) qualifies; a Go
 * slice does not (take the buffers from s2k_host_alloc instead).
 * s2k_ecdsa_verify_batch detects pinned buffers by itself (all four inputs must be pinned). */
void *s2k_host_alloc(size_t bytes);
void s2k_host_free(void *p);
int s2k_host_register(void *p, size_t bytes);
int s2k_host_unregister(void *p);

/* ---- submit / wait: the host-pointer entry points without the wait at their end ------------ */
/* The reference's caller has its data in host memory (secec.PublicKey.Verify, ecdsa.go:171-228) and a cgo shim batches
 * it into s2k_ecdsa_verify_batch, which pays the PCIe transfer and the kernels in series (7.1 ms from pinned, 8.2 ms
 * from pageable memory against 4.9 ms resident per 2^20 signatures).  s2k_ecdsa_verify_batch_submit returns as soon as
 * the batch is enqueued (from page-locked buffers: at once; from pageable ones: when the runtime has staged the copies)
 * and s2k_wait blocks until the verdicts of that ticket are in `valid`.  The context keeps up to FOUR batches in flight on
 * internal child contexts (own workspaces on the same device; the generator tables are shared), in two lanes: even and odd
 * tickets run on two sets of streams beside each other, within a lane the kernels of consecutive tickets follow each other,
 * and a ticket crosses PCIe while its lane's previous ticket computes - a caller that keeps three or four batches
 * submitted sees a little more than the resident rate of a single stream.  A fifth submit first retires the oldest ticket
 * (delivers its verdicts; a later s2k_wait on it returns at once).  s2k_poll is s2k_wait without the blocking: S2K_PENDING while the
 * ticket is in flight.  Same verdicts as s2k_ecdsa_verify_batch, bit for bit; the inputs and `valid` must
 * stay untouched until the ticket has been waited for (or retired).  Submit, wait and the other calls of one context
 * must come from one thread at a time, like all calls on a context; the key-grouping settings are those the context
 * has at submit time.  s2k_wait_all retires everything in flight (oldest first; s2k_ctx_destroy does the same).
 * Each child holds what s2k_ctx_device_bytes reports minus the generator tables. */
typedef uint64_t s2k_ticket;
int s2k_ecdsa_verify_batch_submit(s2k_ctx *ctx, size_t n, const uint8_t *pub_xy, const uint8_t *digest32, const uint8_t *r,
                                  const uint8_t *s, uint32_t flags, uint8_t *valid, s2k_ticket *ticket);
int s2k_wait(s2k_ctx *ctx, s2k_ticket ticket);
int s2k_poll(s2k_ctx *ctx, s2k_ticket ticket);
int s2k_wait_all(s2k_ctx *ctx);
/* Small batches.  A call of up to a few thousand signatures leaves the device empty whatever it runs, and costs the latency
 * of ONE signature's ladder; for batches of up to max_n items (default 3072; 0 = never) s2k_ecdsa_verify_batch,
 * s2k_schnorr_verify_batch and s2k_ecdsa_recover_batch (and their _device / _submit forms) therefore run ladders that spend
 * a whole wavefront on each item (k_verify_row / k_schnorr_row / k_recover_row: the row arithmetic of fe29r.h, complete
 * formulas), and the synchronous host forms move their bytes without DMA transfers (the CPU copies them into a page-locked
 * block the kernels read and write in place): host to host 0.20 instead of 0.74 ms for 1024 ECDSA signatures, 0.19 instead of
 * 0.83 for BIP-340, 0.24 instead of 0.87 for recovery - the reference's own shape is a loop of single PublicKey.Verify calls (secec/ecdsa.go:171; BASELINE config 1
 * verifies 1024).  Same results (tests/test_gpu_round5.py::test_small_batch_*).  A grouping mode set by name
 * (S2K_KEYS_AUTO / S2K_KEYS_ALWAYS) is obeyed at every size; the default (S2K_KEYS_ADAPTIVE) and S2K_KEYS_OFF take these
 * ladders. */
int s2k_ctx_set_small_batch_max(s2k_ctx *ctx, uint32_t max_n);
/* Batches above that threshold and up to max_n items (default 32768; 0 = never) run ladders with FOUR lanes per signature
 * (k_verify_quad / k_schnorr_quad / k_recover_quad, pt29q.h): calls of this size fill neither kind of kernel, and what they cost
 * is the latency of one wave's ladder - half as long this way as with a lane per signature (0.35 instead of 0.6-0.7 ms for
 * 2^12 .. 2^14 ECDSA signatures).  Same results (tests/test_gpu_round5.py). */
int s2k_ctx_set_mid_batch_max(s2k_ctx *ctx, uint32_t max_n);
/* Times of a ticket on the device's clock, for placement diagnostics (s2k_group_member_stats_ex): after
 * s2k_ctx_ticket_timing(ctx, 1) every submitted ticket records three moments on the device's clock: t0 = its first host-to-device
 * copy is about to start, t1 = its last copy has ended, t2 = its verdicts are in host memory.  t0 <= t1 <= t2 by construction
 * (the stream that delivers the verdicts waits for the marker behind the copies), also when several contexts share the
 * device's hardware queues.  s2k_ticket_times gives ms[0] = t1 - t0 (the copies), ms[1] = t2 - t0 (first copy to verdicts), so
 * 0 < ms[0] <= ms[1], for one of the last eight retired tickets (S2K_PENDING and zeros otherwise). */
int s2k_ctx_ticket_timing(s2k_ctx *ctx, int enable);
int s2k_ticket_times(s2k_ctx *ctx, s2k_ticket ticket, double ms[2]);
/* s2k_ecdsa_verify_batch_keyset in the same form: signatures that name their key by its index in a key set of this context
 * (100 bytes per signature cross PCIe instead of 160, and no table is built).  Tickets of all submit entry points share
 * the context's four slots and may be mixed.  The key set has to live until the ticket has been waited for (s2k_keyset_destroy
 * waits for the device first, so destroying it early blocks rather than breaks). */
int s2k_ecdsa_verify_batch_keyset_submit(s2k_ctx *ctx, const s2k_keyset *ks, size_t n, const uint32_t *key_index,
                                         const uint8_t *digest32, const uint8_t *r, const uint8_t *s, uint32_t flags,
                                         uint8_t *valid, s2k_ticket *ticket);
/* s2k_ecdsa_verify_encoded_batch (below) in the same form */
int s2k_ecdsa_verify_encoded_batch_submit(s2k_ctx *ctx, size_t n, const uint8_t *pubs, const uint64_t *pub_off,
                                          const uint8_t *digests, const uint64_t *dig_off, const uint8_t *sigs,
                                          const uint64_t *sig_off, int encoding, size_t digest_len, uint32_t flags,
                                          uint8_t *valid, s2k_ticket *ticket);

/* ---- several devices in one process ------------------------------------------------------------ */
/* north_star: "batches shard trivially across the 8 GPUs of one node".  A Go process cannot be one rank of a
 * torch.distributed job; in ONE process no collective is needed at all.  A GROUP owns one context and one host thread per
 * listed device (a device may be listed more than once: two members then share it, each with its own context).  A batch
 * is cut into contiguous index shards, one per member (signature i belongs to shard i * members / n
 * up to rounding to 256), every member runs s2k_ecdsa_verify_batch_submit / s2k_wait on its shard from its own thread,
 * and the verdicts land in the caller's `valid` at the shard's offset - the "all-gather" is the host array itself.
 * Verdicts are those of s2k_ecdsa_verify_batch on the whole batch, bit for bit (signatures are independent; only the
 * grouping by key is per shard).  s2k_group_ecdsa_verify_batch = submit + wait.  Up to four group batches are in flight
 * per member (a fifth submit blocks until the oldest is done).  Listing a device twice is for tests: two members then run
 * four lanes on it, which is slower than one member's two.  Group calls may come from any ONE thread at a time.
 * s2k_device_count: devices visible to the runtime (0 when there is none). */
typedef struct s2k_group s2k_group;
int s2k_device_count(void);
/* Host topology of a device (topology.cpp; sysfs, S2K_SYSFS_ROOT replaces "/sys"): its PCI bus id ("0000:05:00.0"), the NUMA
 * node it hangs off (-1: unknown, single-node machines included), and the two actions built on it: restrict the CALLING
 * thread to the CPUs of a node that it may run on (returns how many, 0 = nothing changed; never widens the mask), prefer
 * a node for a range of pages (0 = done or nothing to do, -1 = refused).  s2k_topology_*: the parsers, with the sysfs
 * root as an argument (NULL: the default). */
int s2k_device_pci_bus_id(int device, char *out, size_t len /* >= 13 */);
int s2k_device_numa_node(int device);
int s2k_bind_thread_to_node(int node);
int s2k_topology_prefer_node(void *p, size_t bytes, int node);
int s2k_topology_node_count(const char *sysfs_root);
int s2k_topology_numa_node_of_pci(const char *sysfs_root, const char *bus_id);
int s2k_topology_node_cpus(const char *sysfs_root, int node, int *cpus, size_t cap);
int s2k_group_create(const int *devices, size_t n_devices, s2k_group **out);
/* ... with the members' generator table width and context flags as s2k_ctx_create_ex takes them (0, 0 = s2k_group_create) */
int s2k_group_create_ex(const int *devices, size_t n_devices, int gt_bits, uint32_t flags, s2k_group **out);
void s2k_group_destroy(s2k_group *g);
size_t s2k_group_size(const s2k_group *g);
const char *s2k_group_last_error(const s2k_group *g);
int s2k_group_set_key_grouping(s2k_group *g, int mode, uint32_t min_group, uint32_t hash_bits, uint32_t max_tables);
int s2k_group_set_small_batch_max(s2k_group *g, uint32_t max_n);   /* s2k_ctx_set_small_batch_max on every member (a member's shard is what counts as the batch) */
int s2k_group_set_mid_batch_max(s2k_group *g, uint32_t max_n);     /* s2k_ctx_set_mid_batch_max on every member */
int s2k_group_ecdsa_verify_batch(s2k_group *g, size_t n, const uint8_t *pub_xy, const uint8_t *digest32, const uint8_t *r,
                                 const uint8_t *s, uint32_t flags, uint8_t *valid);
int s2k_group_ecdsa_verify_batch_submit(s2k_group *g, size_t n, const uint8_t *pub_xy, const uint8_t *digest32,
                                        const uint8_t *r, const uint8_t *s, uint32_t flags, uint8_t *valid, s2k_ticket *ticket);
int s2k_group_wait(s2k_group *g, s2k_ticket ticket);
/* s2k_ecdsa_verify_encoded_batch (below) across the group: the three blobs are shared, every member takes a contiguous range of
 * ITEMS (its slice of the offset arrays; the offsets stay absolute). */
int s2k_group_ecdsa_verify_encoded_batch(s2k_group *g, size_t n, const uint8_t *pubs, const uint64_t *pub_off,
                                         const uint8_t *digests, const uint64_t *dig_off, const uint8_t *sigs,
                                         const uint64_t *sig_off, int encoding, size_t digest_len, uint32_t flags, uint8_t *valid);
int s2k_group_ecdsa_verify_encoded_batch_submit(s2k_group *g, size_t n, const uint8_t *pubs, const uint64_t *pub_off,
                                                const uint8_t *digests, const uint64_t *dig_off, const uint8_t *sigs,
                                                const uint64_t *sig_off, int encoding, size_t digest_len, uint32_t flags,
                                                uint8_t *valid, s2k_ticket *ticket);
/* stats[m * 4 + 0..3] for member m, of its last finished shard: signatures, first index, milliseconds from the member's
 * submit to its verdicts (host clock), device index. */
int s2k_group_member_stats(s2k_group *g, double *stats /* 4 * members */);
/* The same and more, stats[m * 8 + 0..7]: signatures, first index, host milliseconds, device index, NUMA node of the device
 * (-1 unknown), CPUs the member's thread is bound to (0: not bound), and of the last finished shard [6] the milliseconds of its
 * host-to-device copies and [7] the milliseconds from its first copy to its verdicts, on the device's clock as
 * s2k_ticket_times defines them: 0 < [6] <= [7] for every member, whatever shares its device (the first call switches that
 * timing on: zero until a shard has been submitted after it).  Placement: every member's thread binds itself
 * to the CPUs of its device's NUMA node (sysfs; no-op on one node, when the node is unknown or the cpuset forbids it). */
int s2k_group_member_stats_ex(s2k_group *g, double *stats /* 8 * members */);
/* Blocks until the wide generator tables of every member's device are in (s2k_ctx_gt_wait per member); returns the smallest
 * window width in use.  For benchmarks and services that want their full rate from the first batch. */
int s2k_group_gt_wait(s2k_group *g);
/* Items per member of a batch of n: member i takes [i * size, min(n, (i + 1) * size)). */
size_t s2k_group_shard_size(const s2k_group *g, size_t n);
/* A page-locked array of n items of bytes_per_item bytes laid out for the group: the pages of every member's shard are
 * placed on the NUMA node of that member's device (first touched there; mbind where the kernel allows), so that on a
 * two-socket node no device pulls its shard across the socket link.  One node / unknown nodes: an ordinary pinned block.
 * NULL on failure (s2k_group_last_error).  Freed by s2k_group_host_free or with the group.  The reference keeps its data on
 * the Go heap (secec/ecdsa.go:171-228), which cannot be pinned (s2k_host_register above): a cgo shim fills these instead. */
void *s2k_group_host_alloc(s2k_group *g, size_t bytes_per_item, size_t n);
void s2k_group_host_free(s2k_group *g, void *p);
/* BASELINE configs 3 and 4 across a group (synchronous; the multi-process forms are sharding.py's).  The BIP-340 whole-batch
 * check: every member checks its contiguous shard as one random linear combination (s2k_schnorr_batch_verify_rlc; the
 * members' coefficients are independent: every call mixes fresh operating-system randomness into the seed), *all_valid = 1
 * iff every shard is accepted.  The multiscalar multiplication: every member sums its shard of the terms, the first member
 * adds the partial sums.  Same results as the single-context calls (Point.MultiScalarMultVartime, point_mul_multi.go:73-117: variable time). */
int s2k_group_schnorr_batch_verify_rlc(s2k_group *g, size_t n, const uint8_t *pk /* n*32 */, const uint8_t *msgs,
                                       const uint64_t *msg_offsets, size_t msg_len, const uint8_t *sig /* n*64 */,
                                       const uint8_t *seed32, int *all_valid);
int s2k_group_multi_scalar_mult(s2k_group *g, size_t n, const uint8_t *k /* n*32 */, const uint8_t *points /* n*65 */,
                                uint8_t *out65);
/* Key sets across a group: every member builds the tables of all n_keys keys on its own device (side by side; the
 * tables are replicated per device like the generator tables: s2k_group_keyset_device_bytes per member), and
 * s2k_group_ecdsa_verify_batch_keyset[_submit] shards a batch of (key index, digest, r, s) items like any other group
 * call, each member verifying its shard over its own copy (s2k_ecdsa_verify_batch_keyset_submit).  Same verdicts as
 * s2k_ecdsa_verify_batch_keyset on one device.  The set belongs to its group and is destroyed before it; tickets are those
 * of s2k_group_wait.  What secec.PublicKey caches per key (secec/secec.go:80-85), for a stable key set, on every GPU of the node. */
typedef struct s2k_group_keyset s2k_group_keyset;
int s2k_group_keyset_create(s2k_group *g, size_t n_keys, const uint8_t *pub_xy /* n_keys*64, host */, int layout,
                            s2k_group_keyset **out);
void s2k_group_keyset_destroy(s2k_group_keyset *gks);
size_t s2k_group_keyset_size(const s2k_group_keyset *gks);
int s2k_group_keyset_layout(const s2k_group_keyset *gks);
size_t s2k_group_keyset_device_bytes(const s2k_group_keyset *gks);
int s2k_group_ecdsa_verify_batch_keyset(s2k_group *g, const s2k_group_keyset *gks, size_t n, const uint32_t *key_index,
                                        const uint8_t *digest32, const uint8_t *r, const uint8_t *s, uint32_t flags,
                                        uint8_t *valid);
int s2k_group_ecdsa_verify_batch_keyset_submit(s2k_group *g, const s2k_group_keyset *gks, size_t n, const uint32_t *key_index,
                                               const uint8_t *digest32, const uint8_t *r, const uint8_t *s, uint32_t flags,
                                               uint8_t *valid, s2k_ticket *ticket);

/* Packs valid[n] (0/1 bytes, device) into a bitmap (bit i of byte i/8, LSB first; (n+7)/8
 * bytes, device) and writes the number of valid items to *d_count (uint64, device).  This is
 * the payload of the multi-GPU bitmap all-gather (the count travels behind the bitmap shard). */
int s2k_pack_valid_device(s2k_ctx *ctx, size_t n, const void *d_valid, void *d_bitmap, void *d_count,
                          void *hip_stream);
/* Bytes of the per-signature device workspace (tables, scratch, worklist) for batches of up to n signatures.  It is
 * ONE of the context's buffers: see s2k_ctx_device_bytes for everything a verification call of n signatures holds. */
size_t s2k_ecdsa_workspace_bytes(size_t n);
/* Device memory the context holds once it has verified a batch of n signatures with its current key-grouping
 * settings (s2k_ctx_set_key_grouping): the resident generator tables (40 GiB), the per-signature workspace above, and -
 * with the grouping on, the default, for n >= 256 - the grouping arrays and the per-key table buffer (n / min_group
 * tables of 9 KiB, at most max_tables: 2.4 GB at n = 2^20, 38 GB at 2^24, whether or not keys repeat).  The multi-scalar and BIP-340
 * batch entry points keep a workspace of their own on top (about 1.4 KB per term).  Size HBM by this, not by
 * s2k_ecdsa_workspace_bytes alone, when several contexts share a device. */
size_t s2k_ctx_device_bytes(const s2k_ctx *ctx, size_t n);

/* ---- host-side ingest: the reference's byte-level parsing, in batch --------------------- */
/* ParseASN1Signature (secec/s11n.go:83): strict DER SEQUENCE { r INTEGER, s INTEGER }, r, s in
 * [1, n).  Returns 0 and fills r, s (32-byte big-endian); 1 = malformed ASN.1
 * (errInvalidAsn1Sig); 2 = scalar out of range (errInvalidScalar).  Pure host code. */
int s2k_parse_asn1_signature(const uint8_t *der, size_t len, uint8_t r[32], uint8_t s[32]);
/* ParseCompactSignature (secec/s11n.go:129): [R | S], 64 bytes.  Same return codes. */
int s2k_parse_compact_signature(const uint8_t *sig, size_t len, uint8_t r[32], uint8_t s[32]);
/* bitcoin.IsValidSignatureEncodingBIP0066 (secec/bitcoin/asn1_shitcoin.go:13): 1 / 0. */
int s2k_is_valid_signature_encoding_bip0066(const uint8_t *sig_with_sighash, size_t len);
/* PublicKey.Verify(digest, sig, opts) (secec/ecdsa.go:171) for n encoded items: SEC1 public
 * keys (33 or 65 bytes, as secec.NewPublicKey accepts, secec.go:188), digests, and signatures
 * in `encoding`, each as a concatenation with n+1 byte offsets.  digest_len = 0 means
 * opts == nil (any digest of >= 32 bytes); otherwise digests of another length verify false
 * (ecdsa.go:184-188).  flags: S2K_ECDSA_REJECT_MALLEABLE, S2K_ECDSA_BIP0066,
 * S2K_ECDSA_FORCE_COMPLETE.  The bytes are uploaded as they are; parsing, decompression of
 * compressed keys and verification all run on the device.  valid[i] is the bool the
 * reference's Verify returns; keys it could not even construct give 0.
 * encoding S2K_ENCODING_COMPACT_RECOVERABLE (ecdsa.go:204-205,220-226): the key is recovered from (digest, r, s, v) on
 * the verification ladder (RecoverPublicKey, ecdsa.go:244-282) and compared with the supplied one on the device
 * (PublicKey.Equal: the serialised points, secec.go:121-129); RejectMalleable applies before the recovery (ecdsa.go:212).
 * s2k_ecdsa_verify_encoded_batch_submit: the same without the wait at the end (s2k_wait; see
 * s2k_ecdsa_verify_batch_submit). */
int s2k_ecdsa_verify_encoded_batch(s2k_ctx *ctx, size_t n, const uint8_t *pubs, const uint64_t *pub_off,
                                   const uint8_t *digests, const uint64_t *dig_off, const uint8_t *sigs,
                                   const uint64_t *sig_off, int encoding, size_t digest_len, uint32_t flags,
                                   uint8_t *valid);

/* ---- BIP-340 Schnorr verification (batched) ---------------------------------------- */
/* For each i < n: bitcoin.SchnorrPublicKey.Verify(msg_i, sig_i) for the x-only key pk_i
 * (secec/bitcoin/schnorr.go:221-253): valid[i] = 1 iff pk_i is a valid x-only key
 * (NewSchnorrPublicKey, :257-275), r < p, s < n, and R = s*G - e*P is finite with even y and
 * x(R) == r, e = tagged hash "BIP0340/challenge" of r || pk || msg reduced mod n (:420-449),
 * computed on the device.  Messages: either n strings of msg_len bytes each (msg_offsets ==
 * NULL) or concatenated with msg_offsets[n+1] byte offsets (any lengths, as the reference
 * allows).  flags: S2K_ECDSA_FORCE_COMPLETE only. */
int s2k_schnorr_verify_batch(s2k_ctx *ctx, size_t n, const uint8_t *pk /* n*32 */, const uint8_t *msgs,
                             const uint64_t *msg_offsets, size_t msg_len, const uint8_t *sig /* n*64 */,
                             uint32_t flags, uint8_t *valid /* n */);
int s2k_schnorr_verify_batch_device(s2k_ctx *ctx, size_t n, const void *d_pk, const void *d_msgs,
                                    const void *d_msg_offsets, size_t msg_len, const void *d_sig, uint32_t flags,
                                    void *d_valid, void *hip_stream);
/* The same for signatures that name their key by its index in a key set of this context (s2k_keyset_create[_ex]: X || Y keys).
 * BIP-340's public key is the x coordinate and its point lift_x(x), the one with even y (NewSchnorrPublicKeyFromPoint,
 * schnorr.go:276-300, does that to a point): key k of the set stands for the x-only key X_k whatever the parity of its Y.
 * valid[i] = SchnorrPublicKey(X of keys[key_index[i]]).Verify(msg_i, sig_i); an index outside the set, or a key that is no
 * point of the curve: 0.  Same verdicts as s2k_schnorr_verify_batch on the expanded x-only keys, bit for bit, in every
 * layout of the set; no grouping, key lift or table build per call.  flags: 0. */
int s2k_schnorr_verify_batch_keyset(s2k_ctx *ctx, const s2k_keyset *ks, size_t n, const uint32_t *key_index /* n */,
                                    const uint8_t *msgs, const uint64_t *msg_offsets, size_t msg_len,
                                    const uint8_t *sig /* n*64 */, uint32_t flags, uint8_t *valid);
int s2k_schnorr_verify_batch_keyset_device(s2k_ctx *ctx, const s2k_keyset *ks, size_t n, const void *d_key_index,
                                           const void *d_msgs, const void *d_msg_offsets, size_t msg_len, const void *d_sig,
                                           uint32_t flags, void *d_valid, void *hip_stream);
/* ... and as a ticket of the context's submit / wait slots (s2k_wait / s2k_poll / s2k_wait_all, above) */
int s2k_schnorr_verify_batch_keyset_submit(s2k_ctx *ctx, const s2k_keyset *ks, size_t n, const uint32_t *key_index,
                                           const uint8_t *msgs, const uint64_t *msg_offsets, size_t msg_len,
                                           const uint8_t *sig, uint32_t flags, uint8_t *valid, uint64_t *ticket);

/* Whole-batch BIP-340 verification as ONE multi-scalar multiplication of n + K + 1 points
 * ((sum a_i s_i) G - sum a_i R_i - sum over the K distinct keys P of (sum of a_i e_i over P's signatures) P
 * == infinity, a_0 = 1, a_i = 128 bits of SHA-256(key || i), key = SHA-256(seed32 || 32 bytes of
 * getrandom(2)); the signatures are grouped by key on the device).  *all_valid = 1 iff every
 * signature of the batch verifies (false accept probability 2^-128).  seed32 should be fresh CSPRNG
 * output; because the library mixes in operating-system randomness of its own, a reused or
 * predictable seed does not make the coefficients predictable.  It does not say which signature
 * fails — s2k_schnorr_verify_batch_bisect does.  The
 * reference has single verification only (schnorr.go:221); this is BASELINE config 4. */
/* (Host-pointer forms of the whole-batch calls - this one, s2k_schnorr_verify_batch_bisect, s2k_multi_scalar_mult - are
 * synchronous: transfer, kernels, answer.  Two contexts on two host threads that take whole batches alternately overlap one's
 * transfer with the other's kernels; the library makes the phases of such calls take turns per device, so the two settle at
 * about the resident rate instead of falling into lock step.) */
int s2k_schnorr_batch_verify_rlc(s2k_ctx *ctx, size_t n, const uint8_t *pk, const uint8_t *msgs,
                                 const uint64_t *msg_offsets, size_t msg_len, const uint8_t *sig,
                                 const uint8_t *seed32, int *all_valid);
int s2k_schnorr_batch_verify_rlc_device(s2k_ctx *ctx, size_t n, const void *d_pk, const void *d_msgs,
                                        const void *d_msg_offsets, size_t msg_len, const void *d_sig,
                                        const uint8_t *seed32 /* host */, int *all_valid /* host */, void *hip_stream);

/* Per-signature BIP-340 verdicts at the price of the whole-batch check when everything verifies
 * (the usual case): one combination over the batch; when it is rejected, the failing signatures
 * are located by bisection — the lifted points, challenges and coefficients of the whole batch are
 * kept, a half-range is re-checked as a multiscalar multiplication of its own terms, the other
 * half's error point follows by subtraction, and ranges of <= 2^17 signatures (or everything left,
 * once more than 8 ranges fail on one level) go through s2k_schnorr_verify_batch.  valid[i] is
 * what SchnorrPublicKey.Verify returns for item i (schnorr.go:221-253), up to a false accept
 * probability of 2^-128 per combination.  stats (host, may be NULL): [0] sub-range combinations,
 * [1] signatures verified one by one, [2] levels descended, [3] 1 = bisection abandoned. */
int s2k_schnorr_verify_batch_bisect(s2k_ctx *ctx, size_t n, const uint8_t *pk, const uint8_t *msgs,
                                    const uint64_t *msg_offsets, size_t msg_len, const uint8_t *sig,
                                    const uint8_t *seed32, uint8_t *valid, uint32_t stats[4]);
int s2k_schnorr_verify_batch_bisect_device(s2k_ctx *ctx, size_t n, const void *d_pk, const void *d_msgs,
                                           const void *d_msg_offsets, size_t msg_len, const void *d_sig,
                                           const uint8_t *seed32 /* host */, void *d_valid, uint32_t stats[4] /* host */,
                                           void *hip_stream);

/* ---- group operations (batched; host pointers) -------------------------------------
 * VARIABLE TIME, like everything that runs on the GPU here: these serve the reference's *Vartime
 * code paths (public scalars: verification, recovery, batch checks).  They must NOT be bound to
 * Point.ScalarMult / Point.ScalarBaseMult, which the reference keeps constant-time for ECDH and
 * signing (point_mul_glv.go:257, point_mul_table.go:168): those are served by the s2k_ct_*
 * functions below, on the host CPU. */
/* out[i] = k[i]*G — scalarBaseMultVartime (point_mul_table.go:197) */
int s2k_scalar_base_mult_batch(s2k_ctx *ctx, size_t n, const uint8_t *k /* n*32 */, uint8_t *out /* n*65 */);
/* out[i] = k[i]*P[i] — scalarMultVartimeGLV (point_mul_glv.go:203) */
int s2k_scalar_mult_batch(s2k_ctx *ctx, size_t n, const uint8_t *k, const uint8_t *points /* n*65 */, uint8_t *out);
/* out[i] = u1[i]*G + u2[i]*P[i] — Point.DoubleScalarMultBasepointVartime (point_mul_glv.go:307) */
int s2k_double_scalar_mult_basepoint_batch(s2k_ctx *ctx, size_t n, const uint8_t *u1, const uint8_t *u2,
                                           const uint8_t *points, uint8_t *out);
/* The same two operations with an implementation selector (for the parity tests, which run the
 * reference's vectors through BOTH implementations; the entry points above use S2K_IMPL_FAST):
 *   S2K_IMPL_COMPLETE  the reference's algorithm shape: complete projective formulas on the 8x32
 *                      field (what k_verify_fallback runs for lanes the ladder cannot decide);
 *   S2K_IMPL_FAST      the verification ladder itself (k_verify_fast: 9x29 lazy field, common-Z
 *                      odd-multiple table, signed odd digits of the odd GLV split, Jacobian
 *                      doublings / mixed additions, resident generator tables) with undecided
 *                      lanes re-done by the complete path, exactly as in a verification.
 * u1 == NULL: out[i] = u2[i]*P[i] (scalarMultVartimeGLV).  A record that is neither 0x04||X||Y on
 * the curve nor the identity record gives S2K_ERR_ARG (the reference cannot construct such Points). */
#define S2K_IMPL_COMPLETE 0u
#define S2K_IMPL_FAST 1u
int s2k_double_scalar_mult_basepoint_batch_ex(s2k_ctx *ctx, uint32_t impl, size_t n, const uint8_t *u1,
                                              const uint8_t *u2, const uint8_t *points, uint8_t *out);
/* out[i] = a[i] + b[i] — Point.Add (point.go:62); out[i] = 2*a[i] — Point.Double (point.go:71) */
int s2k_point_add_batch(s2k_ctx *ctx, size_t n, const uint8_t *a, const uint8_t *b, uint8_t *out);
int s2k_point_double_batch(s2k_ctx *ctx, size_t n, const uint8_t *a, uint8_t *out);
/* out = sum_i k[i] * P[i] — Point.MultiScalarMultVartime (point_mul_multi.go:73-117) ONLY: variable time
 * (data-dependent bucket sorts).  The constant-time Point.MultiScalarMult (:25-67) binds to
 * s2k_ct_multi_scalar_mult below, never to this.  n == 0 gives the identity.  Pippenger bucket method with
 * complete additions (the reference uses Straus; same group element).  A malformed point
 * record is S2K_ERR_ARG.  The device form takes device pointers and synchronises the stream
 * before returning (it has to read back the status word). */
int s2k_multi_scalar_mult(s2k_ctx *ctx, size_t n, const uint8_t *k /* n*32 */, const uint8_t *points /* n*65 */,
                          uint8_t *out /* 65 */);
int s2k_multi_scalar_mult_device(s2k_ctx *ctx, size_t n, const void *d_k, const void *d_points, void *d_out65,
                                 void *hip_stream);
/* SEC1 decode of n fixed-size encodings (enc_len = 33: SetCompressedBytes point_s11n.go:140;
 * enc_len = 65: SetUncompressedBytes :178).  ok[i] = 1 and out[i] = point on success,
 * ok[i] = 0 and out[i] = zeros otherwise. */
int s2k_point_decode_batch(s2k_ctx *ctx, size_t n, size_t enc_len, const uint8_t *enc, uint8_t *out, uint8_t *ok);

/* ---- constant-time twins, host CPU (no GPU, no context) ------------------------------------
 * What the reference's secret-handling code calls.  No branch, address or
 * loop count depends on the scalar, the private key or the nonce: masked full-table scans
 * (lookupProjectivePoint / lookupAffinePoint, point_mul_table_ref.go:11-24), arithmetic selects,
 * complete formulas.  Point arguments are public (the peer's key) and are validated like
 * SetUncompressedBytes (point_s11n.go:178); a malformed record is S2K_ERR_ARG. */
/* out = k*P — Point.ScalarMult (point_mul_glv.go:257-303); k is reduced mod n (SetBytes) */
int s2k_ct_scalar_mult(const uint8_t k[32], const uint8_t point65[65], uint8_t out65[65]);
/* out = sum_i k[i]*P[i] — Point.MultiScalarMult (point_mul_multi.go:25-67), the CONSTANT-TIME form: Straus with one
 * 15-entry table per point, masked full-table scans (projectivePointMultTable.SelectAndAdd, point_mul_table.go:34-41),
 * doublings shared by all terms; n == 1 is Point.ScalarMult (:31-33), n == 0 the identity.  The scalars (reduced mod n,
 * SetBytes) are secret; the points and n are public.  This — not s2k_multi_scalar_mult — is what Point.MultiScalarMult
 * binds to; only MultiScalarMultVartime (:73-117) may go to the GPU.  Cost: 15 n complete additions for the tables,
 * 252 doublings, 64 n additions; 1440 n bytes of scratch (S2K_ERR_NOMEM if it cannot be had, S2K_ERR_ARG for a
 * malformed record or n > 2^24). */
int s2k_ct_multi_scalar_mult(size_t n, const uint8_t *k /* n*32 */, const uint8_t *points65 /* n*65 */, uint8_t out65[65]);
/* out = k*G — Point.ScalarBaseMult (point_mul_table.go:168-194) */
int s2k_ct_scalar_base_mult(const uint8_t k[32], uint8_t out65[65]);
/* shared_x = x(d*Q) — PrivateKey.ECDH (secec/secec.go:53-56); d in [1,n), Q a valid public key */
int s2k_ct_ecdh(const uint8_t priv32[32], const uint8_t pub65[65], uint8_t shared_x[32]);
/* The arithmetic of PrivateKey.Sign (secec/ecdsa.go:335-390) for a caller-supplied nonce k:
 * r = x(k*G) mod n, s = (e + r*d)/k mod n normalised to s <= n/2, and the recovery id.  The nonce
 * derivation (ecdsa.go:284-333) stays in the caller.  S2K_ERR_ARG: d or k outside [1,n), or r == 0
 * or s == 0 (draw another nonce, as the reference does). */
int s2k_ct_ecdsa_sign_raw(const uint8_t priv32[32], const uint8_t digest32[32], const uint8_t nonce32[32],
                          uint8_t r32[32], uint8_t s32[32], uint8_t *recovery_id);
/* ---- single operations of Point / Scalar / field.Element: constant time, host CPU ----------------------------------
 * What the reference's Point / Scalar / field.Element METHODS bind to.  One call, one operation, no context, no
 * device: the batched GPU forms further down (s2k_point_add_batch, s2k_fn_op_batch, s2k_fp_op_batch) are variable time and
 * cost a device round trip - they are batch offerings for public data, never the binding of a method that may see a secret
 * (Scalar.Invert of a nonce, ecdsa.go:371).  No branch or address depends on an operand's VALUE, including whether a point is
 * the identity.  Points are 65-byte records, scalars / elements 32-byte canonical big-endian; an operand the reference's type
 * cannot hold (off-curve or non-canonical: its constructors return errors, point.go:192-216, scalar.go:135, field.go:153) is
 * S2K_ERR_ARG.  ctrl follows the reference: 0 selects the first alternative, anything else the second. */
int s2k_ct_point_add(const uint8_t a65[65], const uint8_t b65[65], uint8_t out65[65]);        /* Point.Add, point.go:62 */
int s2k_ct_point_double(const uint8_t a65[65], uint8_t out65[65]);                             /* Point.Double, point.go:73 */
int s2k_ct_point_subtract(const uint8_t a65[65], const uint8_t b65[65], uint8_t out65[65]);   /* Point.Subtract, point.go:83 */
int s2k_ct_point_negate(const uint8_t a65[65], uint8_t out65[65]);                             /* Point.Negate, point.go:89 */
int s2k_ct_point_conditional_negate(const uint8_t a65[65], uint64_t ctrl, uint8_t out65[65]); /* Point.ConditionalNegate, point.go:102 */
int s2k_ct_point_conditional_select(const uint8_t a65[65], const uint8_t b65[65], uint64_t ctrl,
                                    uint8_t out65[65]);                                        /* Point.ConditionalSelect, point.go:115 */
int s2k_ct_point_equal(const uint8_t a65[65], const uint8_t b65[65], uint64_t *out);          /* Point.Equal, point.go:133: 1 / 0 */
int s2k_ct_point_is_identity(const uint8_t a65[65], uint64_t *out);                            /* Point.IsIdentity, point.go:148 */
int s2k_ct_point_is_y_odd(const uint8_t a65[65], uint64_t *out);                               /* Point.IsYOdd, point.go:155 */
/* op: S2K_OP_MUL / SQR / ADD / SUB / NEG / INV (below) - Scalar.Multiply / Square / Add / Subtract / Negate (scalar.go:66-93),
 * Scalar.Invert (scalar_invert.go:11; 0 -> 0).  b32 is read by MUL / ADD / SUB only. */
int s2k_ct_scalar_op(int op, const uint8_t a32[32], const uint8_t b32[32], uint8_t out32[32]);
int s2k_ct_scalar_conditional_select(const uint8_t a32[32], const uint8_t b32[32], uint64_t ctrl, uint8_t out32[32]); /* scalar.go:176 */
int s2k_ct_scalar_conditional_negate(const uint8_t a32[32], uint64_t ctrl, uint8_t out32[32]);                        /* scalar.go:168 */
/* what: 0 Scalar.IsZero (scalar.go:187), 1 Scalar.IsGreaterThanHalfN (:196), 2 Scalar.Equal (:182, reads b32): *out = 1 / 0 */
#define S2K_SCALAR_IS_ZERO 0
#define S2K_SCALAR_IS_GT_HALF_N 1
#define S2K_SCALAR_EQUAL 2
int s2k_ct_scalar_predicate(int what, const uint8_t a32[32], const uint8_t b32[32], uint64_t *out);
/* Scalar.SetBytes (scalar.go:123): out = src mod n; *did_reduce (may be NULL) = 1 iff src >= n.  Accepts any 32 bytes. */
int s2k_ct_scalar_set_bytes(const uint8_t src32[32], uint8_t out32[32], uint64_t *did_reduce);
/* field.Element: S2K_OP_MUL / SQR / ADD / SUB / NEG (internal/field/field.go:61-104), S2K_OP_INV (field_invert.go:11; 0 -> 0),
 * S2K_OP_SQRT (field_sqrt_ratio.go:14: *flag = 1 iff a is a square, out = a^((p+1)/4) then, 0 otherwise; flag required). */
int s2k_ct_fe_op(int op, const uint8_t a32[32], const uint8_t b32[32], uint8_t out32[32], uint64_t *flag);
/* Test instrumentation: field multiplications executed by the calling thread in s2k_ct_* since the
 * previous call of this function.  The count is the same for every scalar (tests/test_ct_cpu.py). */
uint64_t s2k_ct_debug_fe_mul_count(void);

/* ---- field / scalar element operations (batched; host pointers) -------------------- */
/* Element / Scalar methods, for API parity and for the parity tests of the device
 * arithmetic.  Inputs are reduced first (SetBytes semantics, field.go:115, scalar.go:123). */
enum {
  S2K_OP_MUL = 0, /* Multiply (field.go:82, scalar.go:84) */
  S2K_OP_SQR = 1, /* Square */
  S2K_OP_ADD = 2, /* Add */
  S2K_OP_SUB = 3, /* Subtract */
  S2K_OP_NEG = 4, /* Negate */
  S2K_OP_INV = 5, /* Invert (field_invert.go:11, scalar_invert.go:11); 0 -> 0 */
  S2K_OP_SQRT = 6 /* Fp only: Sqrt (field_sqrt_ratio.go:14); flag[i] = root exists */
};
int s2k_fp_op_batch(s2k_ctx *ctx, int op, size_t n, const uint8_t *a, const uint8_t *b, uint8_t *out, uint8_t *flag);
int s2k_fn_op_batch(s2k_ctx *ctx, int op, size_t n, const uint8_t *a, const uint8_t *b, uint8_t *out, uint8_t *flag);
/* GLV split (point_mul_glv.go:59): k == k1 + k2*lambda (mod n), both canonical */
int s2k_fn_split_glv_batch(s2k_ctx *ctx, size_t n, const uint8_t *k, uint8_t *k1, uint8_t *k2);

/* Test access to the arithmetic of the hot kernels (impl must be S2K_IMPL_FAST): the 9x29 lazy
 * field of the verification ladder incl. its fused products, the Jacobian doubling / mixed
 * addition (jacobian29.h) and the complete 9x29 formulas of the multiscalar kernels (pt29.h), on
 * operands put in LAZY form first (same value, unreduced limbs; `lazy` holds a 4-bit code per
 * operand: bits 1:0 = multiples of p added limb-wise, bit 2 = borrow-spread).  in[0..4] are
 * n*32-byte big-endian operands a..e (NULL = unused); results are canonical.  Reference symbols
 * served: fiat Mul/Square/Add/Opp (secp256k1montgomery.go:87,418,750,844), Element.Invert / Sqrt
 * (field_invert.go:11, field_sqrt_ratio.go:14), addMixed / addComplete / doubleComplete
 * (point_projective.go:123,24,208).
 *   MUL a*b | SQR a^2 | MUL_PLUS a*b+c | SQR_PLUS a^2+b | MUL_ADD_MUL a*b+c*d | MUL_ADD_SQR a*b+c^2
 *   ADD a+b | NEGATE -a | HALF a/2 | NORMALIZE a (flag = a == 0) | COND_NEGATE1 (b odd ? -a : a)
 *   INV | SQRT (flag = root exists) | EQ (flag) | MUL_SMALL21 21a | NORMALIZE_WEAK a
 *   JDBL: 2P, JADD: P + (d,e) for P = (a,b) lifted to Jacobian Z = c; out,out2 = affine x,y;
 *         flag = 0 when the incomplete formulas hit an exceptional case (Z3 = 0)
 *   PT29_DBL / PT29_ADD / PT29_ADD_MIXED: complete formulas, P = (a:b:1) scaled by Z = c
 *         (c = 0: the identity), Q = (d,e); flag = 0 when the result is the identity */
enum {
  S2K_HP_MUL = 0, S2K_HP_SQR, S2K_HP_MUL_PLUS, S2K_HP_SQR_PLUS, S2K_HP_MUL_ADD_MUL, S2K_HP_MUL_ADD_SQR,
  S2K_HP_ADD, S2K_HP_NEGATE, S2K_HP_HALF, S2K_HP_NORMALIZE, S2K_HP_COND_NEGATE1, S2K_HP_INV, S2K_HP_SQRT,
  S2K_HP_EQ, S2K_HP_MUL_SMALL21, S2K_HP_NORMALIZE_WEAK, S2K_HP_JDBL, S2K_HP_JADD, S2K_HP_PT29_DBL,
  S2K_HP_PT29_ADD, S2K_HP_PT29_ADD_MIXED,
  S2K_HP_INV_GCD,   /* fe29_inv_gcd: the safegcd inversion mod p of the per-key tables (same values as S2K_HP_INV) */
  S2K_HP_JADD_FULL, /* jpt29_add: P as for JADD, Q = (d, e) lifted to Z2 = c^2 */
  /* pt29q.h: the complete formulas spread over the four lanes of a quad (the serial tail of the multi-scalar
   * multiplication); inputs and outputs as PT29_DBL / PT29_ADD (Q scaled by c as well).  Bits 20.. of `lazy`: how often
   * the operation is chained on its own result (P + Q + Q + ..., 2^k P); 0 means once. */
  S2K_HP_PT29Q_DBL, S2K_HP_PT29Q_ADD,
  /* xyzz29_add_affine (xyzz29.h: the bucket pass's incomplete mixed addition): P = (a, b) lifted to ZZ = c^2, ZZZ = c^3,
   * Q = (d, e); flag = 0 when P and Q share their x (ZZ3 = 0: the piece is re-done with the complete formulas) */
  S2K_HP_XYZZ_ADD,
  /* one round border of the keyed ladder as k_verify_fast<ECDSA_KEYED> runs it: P lifted as for XYZZ_ADD, + Q in XYZZ,
   * XYZZ -> Jacobian, two Jacobian doublings, Jacobian -> XYZZ, + Q in XYZZ, XYZZ -> Jacobian: out = 4 P + 5 Q;
   * flag = 0 when Z ends as 0 (P = +-Q: ZZ = 0 must survive the changes of form and the doublings) */
  S2K_HP_XYZZ_ROUND,
  /* fe29r.h: the field with one limb per lane of a 16-lane DPP row, the complete formulas with one product per row (the
   * serial tail of the multi-scalar multiplication since round 5; ONE WAVE per item).  FER_MUL a*b, FER_MUL_PLUS a*b+c,
   * FER_MUL_ADD_MUL a*b+c*d, FER_SMALL 21a: the item's four rows multiply four different lazy forms of the operands, flag =
   * the rows agree.  PT29R_DBL / PT29R_ADD: inputs, outputs and chaining (bits 20.. of `lazy`) as PT29Q_*; bit 0 of `lazy`:
   * P's y with two units; flag 2 = the rows disagree.  FER_SWAPS (n >= 8): out[0..255] = the lane numbers after
   * v_permlane16_swap (even | odd rows) and v_permlane32_swap (low | high half). */
  S2K_HP_FER_MUL, S2K_HP_FER_MUL_PLUS, S2K_HP_FER_MUL_ADD_MUL, S2K_HP_FER_SMALL, S2K_HP_PT29R_DBL, S2K_HP_PT29R_ADD, S2K_HP_FER_SWAPS
};
int s2k_fp_op_batch_ex(s2k_ctx *ctx, uint32_t impl, int op, uint32_t lazy, size_t n, const uint8_t *const in[5],
                       uint8_t *out, uint8_t *out2, uint8_t *flag);
/* The odd GLV split the verification ladder uses (sc_split_glv_odd): k == (-1)^s1 k1 + (-1)^s2 k2 lambda
 * (mod n) with k1, k2 ODD and below 2^129 (32-byte big-endian magnitudes), signs[i] = s1 | s2 << 1.
 * Serves splitGLV (point_mul_glv.go:59); impl must be S2K_IMPL_FAST. */
int s2k_fn_split_glv_batch_ex(s2k_ctx *ctx, uint32_t impl, size_t n, const uint8_t *k, uint8_t *k1, uint8_t *k2,
                              uint8_t *signs);

/* ---- introspection used by the tests ------------------------------------------------ */
/* Copies generator-table entry T_i[d] (X‖Y big-endian) to out64.  Layout: DESIGN.md §3. */
/* window width (bits) the automatic generator tables of this library aim for (a build-time constant: S2K_GT_BITS); what a
 * context uses at a given moment: s2k_ctx_gt_info.  s2k_debug_gtable_entry reads the table the context uses now. */
int s2k_generator_window_bits(void);
int s2k_debug_gtable_entry(s2k_ctx *ctx, unsigned i, unsigned d, uint8_t *out64);

#ifdef __cplusplus
}
#endif
#endif
