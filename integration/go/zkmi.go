// Package zkmi is the cgo binding of libzkmi.so that a gnark maintainer would add next to
// backend/groth16/bn254 (see INTEGRATION.md).
//
// STATUS: SOURCE ONLY, UNBUILT AND UNTESTED.  There is no Go toolchain in the build image, so this
// file has never been compiled; it documents the binding against include/zkmi.h.  The same C
// entry points are exercised through ctypes by tests/ (every symbol, on the GPU).
//
// Shape: gnark keeps its own frontend, constraint system and solver.  ProveBatch solves each
// witness with gnark (cs.Solve -> W, A, B, C), hands the solved vectors to
// zkmi_prove_witness_batch (quotient, five MSMs, assembly on the GPU) and gets gnark Proof values
// back.  The proving key is described by gnark's own fields (InfinityA / InfinityB, nbPublic), so
// nothing but a gnark ccs, pk and witnesses is needed.
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

	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
	"github.com/consensys/gnark/backend"
	groth16_bn254 "github.com/consensys/gnark/backend/groth16/bn254"
	"github.com/consensys/gnark/backend/witness"
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
	h            *C.zkmi_pk
	nWires       int
	nConstraints int
}

// LoadKey uploads a gnark proving key.  maxBatch sizes the HBM plan (zkmi_pk_desc.max_batch).
//
// cgo pointer rules: the descriptor lives in C memory (C.calloc) so that it may hold pointers to
// Go memory, and every Go slice it points to is pinned for the duration of the call.
func (d *Device) LoadKey(pk *groth16_bn254.ProvingKey, nbPublic, nbConstraints, maxBatch int) (*Key, error) {
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
	desc.n_public = C.uint32_t(nbPublic)
	desc.max_batch = C.uint32_t(maxBatch)

	d.mu.Lock()
	defer d.mu.Unlock()
	var h *C.zkmi_pk
	if rc := C.zkmi_pk_load(d.ctx, desc, &h); rc != 0 {
		return nil, d.lastError()
	}
	return &Key{h: h, nWires: len(pk.InfinityA), nConstraints: nbConstraints}, nil
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

// ProveBatch is groth16.Prove(ccs, pk, w) for a batch of independent witnesses of one circuit:
// gnark's solver on the CPU (it can run in goroutines, one per witness), everything else on the
// GPU.  rs holds the two blinding scalars per proof that gnark samples inside Prove.
func (d *Device) ProveBatch(ccs *cs_bn254.R1CS, key *Key, ws []witness.Witness, rs []fr.Element,
	opts ...backend.ProverOption) ([]groth16_bn254.Proof, error) {
	batch := len(ws)
	if len(rs) != 2*batch {
		return nil, errors.New("rs must hold 2 scalars per proof")
	}
	popt, err := backend.NewProverConfig(opts...)
	if err != nil {
		return nil, err
	}
	nw, nc := key.nWires, key.nConstraints
	wires := make([]fr.Element, batch*nw)
	a := make([]fr.Element, batch*nc)
	b := make([]fr.Element, batch*nc)
	c := make([]fr.Element, batch*nc)
	var wg sync.WaitGroup
	errs := make([]error, batch)
	for i := range ws {
		wg.Add(1)
		go func(i int) {
			defer wg.Done()
			sol, err := ccs.Solve(ws[i], popt.SolverOpts...) // gnark's own solver
			if err != nil {
				errs[i] = err
				return
			}
			s := sol.(*cs_bn254.R1CSSolution)
			copy(wires[i*nw:], s.W)
			copy(a[i*nc:], s.A[:nc])
			copy(b[i*nc:], s.B[:nc])
			copy(c[i*nc:], s.C[:nc])
		}(i)
	}
	wg.Wait()
	for _, e := range errs {
		if e != nil {
			return nil, e
		}
	}
	raw := make([]byte, 256*batch) // Ar | Krs | Bs, the field order of gnark's Proof
	d.mu.Lock()
	rc := C.zkmi_prove_witness_batch(d.ctx, key.h, unsafe.Pointer(&wires[0]), unsafe.Pointer(&a[0]),
		unsafe.Pointer(&b[0]), unsafe.Pointer(&c[0]), C.size_t(nc), C.size_t(batch),
		unsafe.Pointer(&rs[0]), unsafe.Pointer(&raw[0]))
	d.mu.Unlock()
	if rc != 0 {
		return nil, d.lastError()
	}
	proofs := make([]groth16_bn254.Proof, batch)
	for i := range proofs { // G1Affine / G2Affine share the memory image of the 256-byte record
		copy(unsafe.Slice((*byte)(unsafe.Pointer(&proofs[i].Ar)), 64), raw[256*i:])
		copy(unsafe.Slice((*byte)(unsafe.Pointer(&proofs[i].Krs)), 64), raw[256*i+64:])
		copy(unsafe.Slice((*byte)(unsafe.Pointer(&proofs[i].Bs)), 128), raw[256*i+128:])
	}
	return proofs, nil
}
