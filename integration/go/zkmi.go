// Package zkmi is the cgo binding of libzkmi.so that a gnark maintainer would add next to
// backend/groth16/bn254 (see INTEGRATION.md).
//
// STATUS: SOURCE ONLY, UNBUILT AND UNTESTED.  There is no Go toolchain in the build image, so this
// file has never been compiled; it documents the binding against include/zkmi.h.  The same C
// entry points are exercised through ctypes by tests/ (every symbol, on the GPU):
// tests/test_gpu_witness_entry.py drives exactly the call sequence below (pinned batch arrays,
// R1CS upload, wire vectors only, two batches in flight), tests/test_gpu_commitment.py the keys
// with commitments.
//
// Shape: gnark keeps its own frontend, constraint system and solver.  A Prover
//
//   - uploads the proving key once (LoadKey: pk.G1/G2, InfinityA/B, nbPublic, CommitmentKeys),
//   - uploads the R1CS matrices once (LoadR1CS: constraint.R1C terms {CID, VID} + cs.Coefficients),
//     so that a batch ships the WIRE VECTORS ONLY and a, b, c are formed on the GPU,
//   - Submit(witnesses): solves every witness with gnark (cs.Solve) in goroutines, each goroutine
//     copying its solution.W into its row of a page-locked batch array (zkmi_host_alloc), then
//     zkmi_prove_witness_submit -- the DMA engine reads the array in place,
//   - Collect(): zkmi_prove_collect[_ex] of the oldest batch -> []groth16_bn254.Proof.
//
// Submit(k+1) before Collect(k) overlaps batch k+1's CPU solve and PCIe transfer with batch k's
// MSM kernels (two batches in flight, as zkmi_prove_submit).
package zkmi

/*
#cgo CFLAGS:  -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../gnark_crypto_primitives_amd -lzkmi
#include <stdlib.h>
#include "zkmi.h"
*/
import "C"

import (
	"errors"
	"runtime"
	"sync"
	"unsafe"

	curve "github.com/consensys/gnark-crypto/ecc/bn254"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
	"github.com/consensys/gnark/backend"
	groth16_bn254 "github.com/consensys/gnark/backend/groth16/bn254"
	"github.com/consensys/gnark/backend/witness"
	"github.com/consensys/gnark/constraint"
	cs_bn254 "github.com/consensys/gnark/constraint/bn254"
)

// Device owns one zkmi context (one per GPU).  A context is thread-compatible: calls are
// serialised by mu, exactly like the lazily-initialised device handle of gnark's icicle backend.
type Device struct {
	ctx *C.zkmi_ctx
	mu  sync.Mutex
}

func Open(device int) (*Device, error) {
	var ctx *C.zkmi_ctx
	if rc := C.zkmi_init(C.int(device), &ctx); rc != 0 {
		return nil, errors.New("zkmi_init failed: no MI355X visible (there is no CPU fallback)")
	}
	return &Device{ctx: ctx}, nil
}

func (d *Device) Close() { C.zkmi_destroy(d.ctx) }

func (d *Device) lastError() error { return errors.New(C.GoString(C.zkmi_last_error(d.ctx))) }

// Key is a device-resident proving key (MSM tables built once, like icicle's lazy upload).
type Key struct {
	h             *C.zkmi_pk
	nWires        int
	nCommitments  int
	commitWires   []int // CommitmentIndex of every commitment
	privateWires  [][]int
	publicAndComm [][]int
}

// LoadKey uploads a gnark proving key.  maxBatch sizes the HBM plan (zkmi_pk_desc.max_batch);
// tableBudget (bytes, 0 = all free HBM) bounds the MSM tables when the GPU is shared.
//
// cgo pointer rules: the descriptors live in C memory (C.calloc) so that they may hold pointers to
// Go memory, and every Go slice they point to is pinned for the duration of the call.
func (d *Device) LoadKey(pk *groth16_bn254.ProvingKey, ccs *cs_bn254.R1CS, maxBatch int,
	tableBudget uint64) (*Key, error) {
	logN := 0
	for n := pk.Domain.Cardinality; n > 1; n >>= 1 {
		logN++
	}
	infA := boolsToBytes(pk.InfinityA) // Go []bool has no guaranteed C layout: copy to bytes
	infB := boolsToBytes(pk.InfinityB)

	desc := (*C.zkmi_pk_desc)(C.calloc(1, C.size_t(unsafe.Sizeof(C.zkmi_pk_desc{}))))
	defer C.free(unsafe.Pointer(desc))
	var pin runtime.Pinner
	defer pin.Unpin()
	p := func(x unsafe.Pointer) unsafe.Pointer { pin.Pin(x); return x }

	desc.log_n = C.uint32_t(logN)
	desc.n_wires = C.uint32_t(len(pk.InfinityA))
	desc.n_a, desc.n_b = C.uint32_t(len(pk.G1.A)), C.uint32_t(len(pk.G1.B))
	// gnark allocates G1.Z with Cardinality entries and uses n - 1: the library takes the first n - 1
	desc.n_k, desc.n_z = C.uint32_t(len(pk.G1.K)), C.uint32_t(len(pk.G1.Z))
	desc.g1_a, desc.g1_b = p(unsafe.Pointer(&pk.G1.A[0])), p(unsafe.Pointer(&pk.G1.B[0]))
	desc.g1_k, desc.g1_z = p(unsafe.Pointer(&pk.G1.K[0])), p(unsafe.Pointer(&pk.G1.Z[0]))
	desc.g2_b = p(unsafe.Pointer(&pk.G2.B[0]))
	desc.g1_alpha, desc.g1_beta = p(unsafe.Pointer(&pk.G1.Alpha)), p(unsafe.Pointer(&pk.G1.Beta))
	desc.g1_delta = p(unsafe.Pointer(&pk.G1.Delta))
	desc.g2_beta, desc.g2_delta = p(unsafe.Pointer(&pk.G2.Beta)), p(unsafe.Pointer(&pk.G2.Delta))
	// a_wire = b_wire = k_wire = NULL: the library derives them from gnark's own fields
	desc.infinity_a = (*C.uint8_t)(p(unsafe.Pointer(&infA[0])))
	desc.infinity_b = (*C.uint8_t)(p(unsafe.Pointer(&infB[0])))
	desc.n_public = C.uint32_t(ccs.GetNbPublicVariables())
	desc.max_batch = C.uint32_t(maxBatch)
	desc.table_budget_bytes = C.uint64_t(tableBudget)

	key := &Key{nWires: len(pk.InfinityA)}
	// Groth16 commitment extension: pk.CommitmentKeys + r1cs.CommitmentInfo
	if info, ok := ccs.CommitmentInfo.(constraint.Groth16Commitments); ok && len(info) > 0 {
		n := len(info)
		cds := (*[1 << 10]C.zkmi_commitment_desc)(C.calloc(C.size_t(n),
			C.size_t(unsafe.Sizeof(C.zkmi_commitment_desc{}))))[:n:n]
		defer C.free(unsafe.Pointer(&cds[0]))
		for i := range info {
			priv := toU32(info[i].PrivateCommitted)
			hashed := toU32(info[i].PublicAndCommitmentCommitted)
			ck := &pk.CommitmentKeys[i]
			cds[i].n_private = C.uint32_t(len(priv))
			cds[i].n_hashed = C.uint32_t(len(hashed))
			cds[i].commitment_wire = C.uint32_t(info[i].CommitmentIndex)
			if len(priv) > 0 {
				cds[i].private_wires = (*C.uint32_t)(p(unsafe.Pointer(&priv[0])))
				cds[i].basis = p(unsafe.Pointer(&ck.Basis[0]))
				cds[i].basis_exp_sigma = p(unsafe.Pointer(&ck.BasisExpSigma[0]))
			}
			if len(hashed) > 0 {
				cds[i].hashed_wires = (*C.uint32_t)(p(unsafe.Pointer(&hashed[0])))
			}
			key.commitWires = append(key.commitWires, info[i].CommitmentIndex)
		}
		desc.n_commitments = C.uint32_t(n)
		desc.commitments = &cds[0]
		key.nCommitments = n
	}

	d.mu.Lock()
	defer d.mu.Unlock()
	if rc := C.zkmi_pk_load(d.ctx, desc, &key.h); rc != 0 {
		return nil, d.lastError()
	}
	return key, nil
}

// R1CS is the device-resident copy of the constraint matrices (zkmi_r1cs_load).
type R1CS struct {
	h            *C.zkmi_r1cs
	nConstraints int
}

// LoadR1CS flattens ccs.GetR1Cs() into three term arrays {CID, VID} with row offsets -- gnark's own
// representation (constraint.Term) -- next to cs.Coefficients.
func (d *Device) LoadR1CS(ccs *cs_bn254.R1CS) (*R1CS, error) {
	r1cs := ccs.GetR1Cs()
	var ptr [3][]uint32
	var terms [3][]C.zkmi_term
	for s := 0; s < 3; s++ {
		ptr[s] = make([]uint32, 1, len(r1cs)+1)
	}
	for _, c := range r1cs {
		for s, le := range [3]constraint.LinearExpression{c.L, c.R, c.O} {
			for _, t := range le {
				terms[s] = append(terms[s], C.zkmi_term{coeff: C.uint32_t(t.CID), wire: C.uint32_t(t.VID)})
			}
			ptr[s] = append(ptr[s], uint32(len(terms[s])))
		}
	}
	desc := (*C.zkmi_r1cs_desc)(C.calloc(1, C.size_t(unsafe.Sizeof(C.zkmi_r1cs_desc{}))))
	defer C.free(unsafe.Pointer(desc))
	var pin runtime.Pinner
	defer pin.Unpin()
	p := func(x unsafe.Pointer) unsafe.Pointer { pin.Pin(x); return x }
	desc.n_wires = C.uint32_t(ccs.GetNbPublicVariables() + ccs.GetNbSecretVariables() + ccs.GetNbInternalVariables())
	desc.n_constraints = C.uint32_t(len(r1cs))
	desc.n_coeffs = C.uint32_t(len(ccs.Coefficients))
	desc.coeffs = p(unsafe.Pointer(&ccs.Coefficients[0]))
	desc.l_ptr, desc.l_terms = (*C.uint32_t)(p(unsafe.Pointer(&ptr[0][0]))), (*C.zkmi_term)(p(unsafe.Pointer(&terms[0][0])))
	desc.r_ptr, desc.r_terms = (*C.uint32_t)(p(unsafe.Pointer(&ptr[1][0]))), (*C.zkmi_term)(p(unsafe.Pointer(&terms[1][0])))
	desc.o_ptr, desc.o_terms = (*C.uint32_t)(p(unsafe.Pointer(&ptr[2][0]))), (*C.zkmi_term)(p(unsafe.Pointer(&terms[2][0])))
	d.mu.Lock()
	defer d.mu.Unlock()
	var h *C.zkmi_r1cs
	if rc := C.zkmi_r1cs_load(d.ctx, desc, &h); rc != 0 {
		return nil, d.lastError()
	}
	return &R1CS{h: h, nConstraints: len(r1cs)}, nil
}

// Prover pipelines batches of one circuit: two page-locked wire arrays, two batches in flight.
type Prover struct {
	d     *Device
	ccs   *cs_bn254.R1CS
	key   *Key
	r1cs  *R1CS
	wires [2][]fr.Element // views of zkmi_host_alloc memory, maxBatch x nWires
	raw   [2]unsafe.Pointer
	slot  int
	sizes []int
}

func (d *Device) NewProver(ccs *cs_bn254.R1CS, key *Key, r1cs *R1CS, maxBatch int) (*Prover, error) {
	pr := &Prover{d: d, ccs: ccs, key: key, r1cs: r1cs}
	for i := range pr.raw {
		n := maxBatch * key.nWires
		pr.raw[i] = C.zkmi_host_alloc(d.ctx, C.size_t(n*fr.Bytes))
		if pr.raw[i] == nil {
			return nil, d.lastError()
		}
		pr.wires[i] = unsafe.Slice((*fr.Element)(pr.raw[i]), n) // Go view of C (pinned) memory
	}
	return pr, nil
}

func (pr *Prover) Close() {
	for _, p := range pr.raw {
		C.zkmi_host_free(pr.d.ctx, p)
	}
}

// Submit solves the witnesses with gnark's own solver (hints, commitment hints included: gnark
// computes the commitment wires itself) straight into the pinned batch array and queues the batch.
// rs holds the two blinding scalars per proof that gnark samples inside Prove.
func (pr *Prover) Submit(ws []witness.Witness, rs []fr.Element, opts ...backend.ProverOption) error {
	batch, nw := len(ws), pr.key.nWires
	if len(rs) != 2*batch {
		return errors.New("rs must hold 2 scalars per proof")
	}
	popt, err := backend.NewProverConfig(opts...)
	if err != nil {
		return err
	}
	dst := pr.wires[pr.slot]
	var wg sync.WaitGroup
	errs := make([]error, batch)
	for i := range ws {
		wg.Add(1)
		go func(i int) {
			defer wg.Done()
			sol, err := pr.ccs.Solve(ws[i], popt.SolverOpts...) // gnark's own solver
			if err != nil {
				errs[i] = err
				return
			}
			copy(dst[i*nw:(i+1)*nw], sol.(*cs_bn254.R1CSSolution).W)
		}(i)
	}
	wg.Wait()
	for _, e := range errs {
		if e != nil {
			return e
		}
	}
	var pin runtime.Pinner
	pin.Pin(&rs[0])
	defer pin.Unpin() // rs is pageable: consumed before the call returns
	pr.d.mu.Lock()
	rc := C.zkmi_prove_witness_submit(pr.d.ctx, pr.key.h, pr.r1cs.h, pr.raw[pr.slot], nil, nil, nil,
		0, C.size_t(batch), unsafe.Pointer(&rs[0]))
	pr.d.mu.Unlock()
	if rc != 0 {
		return pr.d.lastError()
	}
	pr.slot ^= 1
	pr.sizes = append(pr.sizes, batch)
	return nil
}

// Collect returns the proofs of the oldest submitted batch (gnark Proof values, Commitments and
// CommitmentPok filled in for keys with commitments) and the per-proof status (0, or
// ZKMI_ERR_UNSATISFIED when the device's a.b = c check failed).
func (pr *Prover) Collect() ([]groth16_bn254.Proof, []int32, error) {
	batch := pr.sizes[0]
	pr.sizes = pr.sizes[1:]
	raw := make([]byte, 256*batch) // Ar | Krs | Bs, the field order of gnark's Proof
	status := make([]int32, batch)
	nc := pr.key.nCommitments
	coms := make([]curve.G1Affine, batch*(nc+1))
	var pin runtime.Pinner
	pin.Pin(&raw[0])
	pin.Pin(&status[0])
	pin.Pin(&coms[0])
	defer pin.Unpin()
	pr.d.mu.Lock()
	var rc C.int
	if nc > 0 {
		rc = C.zkmi_prove_collect_ex(pr.d.ctx, unsafe.Pointer(&raw[0]), (*C.int32_t)(unsafe.Pointer(&status[0])),
			unsafe.Pointer(&coms[0]))
	} else {
		rc = C.zkmi_prove_collect(pr.d.ctx, unsafe.Pointer(&raw[0]), (*C.int32_t)(unsafe.Pointer(&status[0])))
	}
	pr.d.mu.Unlock()
	if rc != 0 {
		return nil, nil, pr.d.lastError()
	}
	proofs := make([]groth16_bn254.Proof, batch)
	for i := range proofs { // G1Affine / G2Affine share the memory image of the 256-byte record
		copy(unsafe.Slice((*byte)(unsafe.Pointer(&proofs[i].Ar)), 64), raw[256*i:])
		copy(unsafe.Slice((*byte)(unsafe.Pointer(&proofs[i].Krs)), 64), raw[256*i+64:])
		copy(unsafe.Slice((*byte)(unsafe.Pointer(&proofs[i].Bs)), 128), raw[256*i+128:])
		if nc > 0 {
			proofs[i].Commitments = append([]curve.G1Affine(nil), coms[i*(nc+1):i*(nc+1)+nc]...)
			proofs[i].CommitmentPok = coms[i*(nc+1)+nc]
		}
	}
	return proofs, status, nil
}

func boolsToBytes(b []bool) []byte {
	out := make([]byte, len(b))
	for i, v := range b {
		if v {
			out[i] = 1
		}
	}
	return out
}

func toU32(x []int) []uint32 {
	out := make([]uint32, len(x))
	for i, v := range x {
		out[i] = uint32(v)
	}
	return out
}
