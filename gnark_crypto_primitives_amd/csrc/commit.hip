// Groth16 commitment extension of the prover (gnark api.Commit; backend/groth16/bn254/prove.go with
// constraint.Groth16Commitments; gnark-crypto ecc/bn254/fr/pedersen [UPSTREAM-RECALL, SURVEY.md §3.2
// step 6]).  In the reference it is reached through uints.New -> rangecheck
// (utils/uints.go:14-28 <- ecc/secp256k1/ecdsa/address.go:14-40).
//
// For every commitment i of the key:
//   D_i   = sum_j w[private_i[j]] * Basis_i[j]                      (MSM, device, second stream)
//   w[commitment_wire_i] = hash_to_field(D_i.Marshal() || hashed wire values, "bsb22-commitment")
//                                                                     (host: SHA-256 of 64 + 32 k bytes)
//   PoK   = sum_i ch^i * sum_j w[private_i[j]] * BasisExpSigma_i[j]   (MSMs, device, main stream)
//   ch    = hash_to_field(commitment wire values, "G16-BSB22")
// The witness program of such a circuit stops at a COMMIT row (zkmi_cs.commit_rows): the submit
// runs the solver up to it, commits, hashes, writes the challenge into the value file and resumes.
// hash_to_field = RFC 9380 expand_message_xmd over SHA-256, 48 bytes per element (gnark-crypto
// fr.Hash).  There is no CPU fallback: the host only hashes.
#include <cstring>

#include "zkmi_internal.h"
#include "ff29.h"
#include "sha256.h"

using namespace zk;

namespace {

// plain 8 x u32 integer <-> 32 big-endian bytes
void be32(uint8_t out[32], const uint32_t v[8]) {
  for (int i = 0; i < 8; i++) {
    const uint32_t x = v[7 - i];
    out[4 * i] = (uint8_t)(x >> 24);
    out[4 * i + 1] = (uint8_t)(x >> 16);
    out[4 * i + 2] = (uint8_t)(x >> 8);
    out[4 * i + 3] = (uint8_t)x;
  }
}

// fr.Hash(msg, dst, 1)[0] in gnark's Montgomery image
Fr hash_to_fr(const uint8_t* msg, size_t len, const char* dst) {
  const size_t dl = strlen(dst);
  uint8_t dst_prime[64], b0[32], b1[32], b2[32];
  memcpy(dst_prime, dst, dl);
  dst_prime[dl] = (uint8_t)dl;
  const uint8_t zpad[64] = {0}, lib[3] = {0, 48, 0}, one = 1, two = 2;
  Sha256 s0;
  s0.update(zpad, 64);
  s0.update(msg, len);
  s0.update(lib, 3);
  s0.update(dst_prime, dl + 1);
  s0.final(b0);
  Sha256 s1;
  s1.update(b0, 32);
  s1.update(&one, 1);
  s1.update(dst_prime, dl + 1);
  s1.final(b1);
  uint8_t x[32];
  for (int i = 0; i < 32; i++) x[i] = b0[i] ^ b1[i];
  Sha256 s2;
  s2.update(x, 32);
  s2.update(&two, 1);
  s2.update(dst_prime, dl + 1);
  s2.final(b2);
  // u = b1 || b2[:16] big-endian = hi (16 bytes) * 2^256 + lo (32 bytes)
  Fr hi = Fr::zero(), lo = Fr::zero();
  for (int i = 0; i < 16; i++) hi.v[3 - i / 4] |= (uint32_t)b1[i] << (24 - 8 * (i % 4));
  uint8_t lob[32];
  memcpy(lob, b1 + 16, 16);
  memcpy(lob + 16, b2, 16);
  for (int i = 0; i < 32; i++) lo.v[7 - i / 4] |= (uint32_t)lob[i] << (24 - 8 * (i % 4));
  // lo < 2^256 < 6 r: bring it below r (add() and mul() expect reduced operands)
  for (;;) {
    bool ge = true;
    for (int j = 7; j >= 0; j--) {
      if (lo.v[j] != FrParams::p(j)) {
        ge = lo.v[j] > FrParams::p(j);
        break;
      }
    }
    if (!ge) break;
    int64_t br = 0;
    for (int j = 0; j < 8; j++) {
      const int64_t t = (int64_t)lo.v[j] - (int64_t)FrParams::p(j) + br;
      lo.v[j] = (uint32_t)t;
      br = t >> 32;
    }
  }
  const Fr plain = add(to_mont(hi), lo);   // to_mont(hi) = hi * 2^256 mod r as a plain integer
  return to_mont(plain);
}

// value-file domains: gnark's image (x 2^256) or the solver's F domain (x 2^261)
Fr k261_plain() {
  Fr k;
  for (int i = 0; i < 8; i++) k.v[i] = Fr29Params::k261(i);
  return k;
}
Fr km5_plain() {   // 2^-5 mod r
  Fr t = Fr::zero();
  t.v[0] = 32;
  return from_mont(inverse(to_mont(t)));
}
Fr domain_to_plain(const Fr& v, bool f) { return f ? mul(v, km5_plain()) : from_mont(v); }
Fr mont_to_domain(const Fr& m, bool f) { return f ? mul(m, k261_plain()) : m; }

// row `row` of a batch-inner matrix <-> a host vector of Bp elements
void row_pack(std::vector<uint4>& out, const std::vector<Fr>& vals, size_t Bp) {
  out.assign(2 * Bp, make_uint4(0, 0, 0, 0));
  for (size_t p = 0; p < vals.size(); p++) {
    const Fr& x = vals[p];
    out[p] = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    out[Bp + p] = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
  }
}
Fr row_get(const std::vector<uint4>& row, size_t p, size_t Bp) {
  Fr r;
  const uint4 lo = row[p], hi = row[Bp + p];
  r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
  r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
  return r;
}

int window_bits_for(size_t n) { return n <= 2048 ? 8 : n <= 32768 ? 6 : 4; }

// scalars of commitment i scaled by the proof's folding power: tmp[j][p] = w[priv[j]][p] * ch[p]
__global__ __launch_bounds__(256) void pok_scale_kernel(const Fr* __restrict__ slots,
                                                        const uint32_t* __restrict__ priv,
                                                        const Fr* __restrict__ ch, Fr* __restrict__ out,
                                                        uint32_t n, size_t Bp, int f_domain) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= Bp) return;
  const Fr c = bi_ld(ch, 0, p, Bp);
  for (uint32_t j = blockIdx.y; j < n; j += gridDim.y) {
    const Fr v = bi_ld(slots, priv[j], p, Bp);
    Fr r = mul(v, c);   // . 2^-256
    if (f_domain) {     // F-domain product is . 2^-261: one more factor 2^-5 = (2^251) . 2^-256
      Fr k = Fr::zero();
      k.v[7] = 1u << 27;
      r = mul(r, k);
    }
    bi_st(out, j, p, Bp, r);
  }
}
__global__ __launch_bounds__(64) void xyzz_add_kernel(G1XYZZ* acc, const G1XYZZ* b, size_t n, int first) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (first) {
    acc[i] = b[i];
  } else {
    G1XYZZ a = acc[i];
    padd(a, b[i]);
    acc[i] = a;
  }
}

}  // namespace

namespace zk {

void commit_keys_free(zkmi_ctx* ctx, zkmi_pk* pk) {
  for (auto& ck : pk->commits) {
    zkmi_msm_bases_free(ctx, ck.basis);
    zkmi_msm_bases_free(ctx, ck.sigma);
    if (ck.private_dev) hipFree(ck.private_dev);
  }
  pk->commits.clear();
}

int commit_keys_load(zkmi_ctx* ctx, const zkmi_pk_desc* d, zkmi_pk* pk) {
  if (d->n_commitments == 0) return ZKMI_OK;
  if (!d->commitments || d->n_commitments > 64) {
    ctx->err = "pk: n_commitments > 0 needs a commitments array (at most 64)";
    return ZKMI_ERR_ARG;
  }
  pk->commits.resize(d->n_commitments);
  for (uint32_t i = 0; i < d->n_commitments; i++) {
    const zkmi_commitment_desc& c = d->commitments[i];
    zkmi_commit_key& ck = pk->commits[i];
    ck.n_private = c.n_private;
    ck.n_hashed = c.n_hashed;
    ck.wire = c.commitment_wire;
    std::vector<uint32_t> priv(c.n_private);
    ck.hashed.resize(c.n_hashed);
    if ((c.n_private && (!c.private_wires || !c.basis || !c.basis_exp_sigma ||
                         hipMemcpy(priv.data(), c.private_wires, (size_t)c.n_private * 4,
                                   hipMemcpyDefault) != hipSuccess)) ||
        (c.n_hashed && (!c.hashed_wires || hipMemcpy(ck.hashed.data(), c.hashed_wires,
                                                     (size_t)c.n_hashed * 4, hipMemcpyDefault) != hipSuccess))) {
      (void)hipGetLastError();
      ctx->err = "pk: cannot read commitment " + std::to_string(i);
      return ZKMI_ERR_ARG;
    }
    // every wire index is checked here, before a kernel uses it as a row number
    bool ok = ck.wire < d->n_wires && ck.wire != 0;
    for (uint32_t w : priv) ok = ok && w < d->n_wires && w != 0;
    for (uint32_t w : ck.hashed) ok = ok && w < d->n_wires && w != 0;
    if (!ok) {
      ctx->err = "pk: commitment " + std::to_string(i) + " holds a wire index out of range";
      return ZKMI_ERR_ARG;
    }
    if (c.n_private) {
      ZK_HIP(hipMalloc((void**)&ck.private_dev, (size_t)c.n_private * 4));
      ZK_HIP(hipMemcpy(ck.private_dev, priv.data(), (size_t)c.n_private * 4, hipMemcpyHostToDevice));
    }
    if (c.n_private == 0) continue;
    int rc;
    const int wb = window_bits_for(c.n_private);
    if ((rc = zkmi_msm_bases_load(ctx, 1, c.basis, c.n_private, wb, &ck.basis)) ||
        (rc = zkmi_msm_bases_load(ctx, 1, c.basis_exp_sigma, c.n_private, wb, &ck.sigma)))
      return rc;
    ck.basis->side = 1;
  }
  return ZKMI_OK;
}

// buffers of a set for a key with commitments: points (n x Bp affine), folding powers (n rows),
// proof of knowledge (Bp XYZZ + Bp affine)
static int commit_buffers(zkmi_ctx* ctx, zkmi_ctx::ProveSet& S, int si) {
  const size_t n = S.pk->commits.size(), Bp = S.Bp;
  void* base;
  // scratch slots 19 (set 0) / 20 (set 1): owned by the set, like its value file
  int rc = ensure_scratch(ctx, si == 0 ? 19 : 20, n * Bp * 64 + n * Bp * 32 + Bp * (128 + 64) + Bp * 128,
                          &base);
  if (rc) return rc;
  S.commit_pts = base;
  S.commit_ch = (char*)base + n * Bp * 64;
  S.commit_pok = (char*)S.commit_ch + n * Bp * 32;
  S.commit_acc = (char*)S.commit_pok + Bp * (128 + 64);
  return ZKMI_OK;
}

int commit_phase(zkmi_ctx* ctx, zkmi_ctx::ProveSet& S, uint32_t index, bool write_wire) {
  const zkmi_pk* pk = S.pk;
  const size_t n = pk->commits.size(), Bp = S.Bp, batch = S.batch;
  const zkmi_commit_key& ck = pk->commits[index];
  int rc;
  if (index == 0) {
    if ((rc = commit_buffers(ctx, S, (int)(&S - ctx->sets)))) return rc;
    S.commit_host.assign(n * batch, Fr::zero());
  }
  // D = <w[private], Basis>: the value file's wire rows are the scalars
  void* acc = S.commit_acc;
  G1Affine* pts = (G1Affine*)S.commit_pts + index * Bp;
  if (ck.n_private == 0) {   // nothing private committed: the commitment is the point at infinity
    ZK_HIP(hipMemsetAsync(pts, 0, Bp * 64, ctx->stream));
  } else if ((rc = msm_run(ctx, ck.basis, (const Fr*)S.slots, ck.private_dev, Bp, acc, S.f_domain)) ||
             (rc = xyzz_to_affine(ctx, 1, acc, pts, Bp))) {
    return rc;
  }
  if (!write_wire && n == 1) return ZKMI_OK;   // solved witness, one commitment: nothing to hash
  // ---- host: the commitment wire's value.  Solver path: hash of the commitment and of the hashed
  // wires' values.  Witness path: the caller's solver already did that; read the wire back.
  std::vector<uint4> row(2 * Bp);
  const size_t row_bytes = 2 * Bp * 16;
  auto row_ptr = [&](uint32_t w) { return (const char*)S.slots + (size_t)w * row_bytes; };
  Fr* host = S.commit_host.data() + (size_t)index * batch;   // plain integers
  if (!write_wire) {
    ZK_HIP(hipMemcpyAsync(row.data(), row_ptr(ck.wire), row_bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
    for (size_t p = 0; p < batch; p++) host[p] = domain_to_plain(row_get(row, p, Bp), S.f_domain);
    return ZKMI_OK;
  }
  std::vector<G1Affine> hpts(batch);
  std::vector<std::vector<uint4>> hrows(ck.n_hashed, std::vector<uint4>(2 * Bp));
  ZK_HIP(hipMemcpyAsync(hpts.data(), pts, batch * 64, hipMemcpyDeviceToHost, ctx->stream));
  for (uint32_t j = 0; j < ck.n_hashed; j++)
    ZK_HIP(hipMemcpyAsync(hrows[j].data(), row_ptr(ck.hashed[j]), row_bytes, hipMemcpyDeviceToHost,
                          ctx->stream));
  ZK_HIP(hipStreamSynchronize(ctx->stream));
  std::vector<uint8_t> msg(64 + 32 * (size_t)ck.n_hashed);
  std::vector<Fr> vals(batch);
  for (size_t p = 0; p < batch; p++) {
    const G1Affine& d = hpts[p];
    if (d.x.is_zero() && d.y.is_zero()) {   // G1Affine.Marshal() of the point at infinity
      memset(msg.data(), 0, 64);
      msg[0] = 0x40;
    } else {
      be32(msg.data(), from_mont(d.x).v);
      be32(msg.data() + 32, from_mont(d.y).v);
    }
    for (uint32_t j = 0; j < ck.n_hashed; j++)
      be32(msg.data() + 64 + 32 * j, domain_to_plain(row_get(hrows[j], p, Bp), S.f_domain).v);
    const Fr c = hash_to_fr(msg.data(), msg.size(), "bsb22-commitment");
    host[p] = from_mont(c);
    vals[p] = mont_to_domain(c, S.f_domain);
  }
  row_pack(row, vals, Bp);
  ZK_HIP(hipMemcpyAsync((void*)row_ptr(ck.wire), row.data(), row_bytes, hipMemcpyHostToDevice, ctx->stream));
  ZK_HIP(hipStreamSynchronize(ctx->stream));   // `row` is a local
  return ZKMI_OK;
}

int commit_finish_submit(zkmi_ctx* ctx, zkmi_ctx::ProveSet& S) {
  const size_t n = S.pk->commits.size(), Bp = S.Bp, batch = S.batch;
  if (n <= 1) return ZKMI_OK;
  // folding challenge of the proofs of knowledge and its powers, per proof, in the value file's domain
  std::vector<std::vector<Fr>> pw(n, std::vector<Fr>(batch));
  std::vector<uint8_t> ser(32 * n);
  for (size_t p = 0; p < batch; p++) {
    for (size_t i = 0; i < n; i++) be32(ser.data() + 32 * i, S.commit_host[i * batch + p].v);
    const Fr ch = hash_to_fr(ser.data(), ser.size(), "G16-BSB22");
    Fr cur = Fr::one();
    for (size_t i = 0; i < n; i++) {
      pw[i][p] = mont_to_domain(cur, S.f_domain);
      cur = mul(cur, ch);
    }
  }
  std::vector<uint4> row;
  for (size_t i = 0; i < n; i++) {
    row_pack(row, pw[i], Bp);
    ZK_HIP(hipMemcpyAsync((char*)S.commit_ch + i * Bp * 32, row.data(), 2 * Bp * 16,
                          hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(hipStreamSynchronize(ctx->stream));
  }
  return ZKMI_OK;
}

int commit_pok(zkmi_ctx* ctx, zkmi_ctx::ProveSet& S) {
  const zkmi_pk* pk = S.pk;
  const size_t n = pk->commits.size(), Bp = S.Bp;
  if (n == 0) return ZKMI_OK;
  int rc;
  G1XYZZ* pok = (G1XYZZ*)S.commit_pok;
  G1Affine* pok_aff = (G1Affine*)((char*)S.commit_pok + Bp * 128);
  if (n == 1 && pk->commits[0].n_private) {
    const zkmi_commit_key& ck = pk->commits[0];
    if ((rc = msm_run(ctx, ck.sigma, (const Fr*)S.slots, ck.private_dev, Bp, pok, S.f_domain))) return rc;
  } else {
    size_t mx = 1;
    for (auto& ck : pk->commits) mx = std::max<size_t>(mx, ck.n_private);
    void *tmp, *part;
    if ((rc = ensure_scratch(ctx, 7, mx * Bp * 32 + Bp * 128, &tmp))) return rc;
    part = (char*)tmp + mx * Bp * 32;
    for (size_t i = 0; i < n; i++) {
      const zkmi_commit_key& ck = pk->commits[i];
      if (ck.n_private == 0) {
        if (i == 0) {   // accumulator starts at the identity
          std::vector<G1XYZZ> inf(Bp, G1XYZZ::inf());
          ZK_HIP(hipMemcpyAsync(pok, inf.data(), Bp * 128, hipMemcpyHostToDevice, ctx->stream));
          ZK_HIP(hipStreamSynchronize(ctx->stream));
        }
        continue;
      }
      hipLaunchKernelGGL(pok_scale_kernel, dim3((unsigned)((Bp + 255) / 256), (unsigned)std::min<uint32_t>(ck.n_private, 1024)),
                           dim3(256), 0, ctx->stream, (const Fr*)S.slots, ck.private_dev,
                           (const Fr*)((char*)S.commit_ch + i * Bp * 32), (Fr*)tmp, ck.n_private, Bp,
                           S.f_domain ? 1 : 0);
      if ((rc = msm_run(ctx, ck.sigma, (const Fr*)tmp, nullptr, Bp, part, S.f_domain))) return rc;
      hipLaunchKernelGGL(xyzz_add_kernel, dim3((unsigned)(Bp / 64)), dim3(64), 0, ctx->stream, pok,
                         (const G1XYZZ*)part, Bp, i == 0 ? 1 : 0);
    }
  }
  if ((rc = xyzz_to_affine(ctx, 1, pok, pok_aff, Bp))) return rc;
  ZK_HIP(hipGetLastError());
  return ZKMI_OK;
}

}  // namespace zk
