// Package zkmi is the cgo binding of libzkmi.so that a gnark maintainer would add next to
// backend/groth16/bn254 (see INTEGRATION.md).  SOURCE ONLY: there is no Go toolchain in the build
// image, this file has never been compiled.  The C-ABI it binds is ../../include/zkmi.h.
package zkmi

/*
#cgo CFLAGS:  -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../gnark_crypto_primitives_amd -lzkmi
#include "zkmi.h"
*/
import "C"

import (
	"errors"
	"sync"
	"unsafe"

	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
	groth16_bn254 "github.com/consensys/gnark/backend/groth16/bn254"
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

// LoadKey uploads a gnark proving key once.  aWire/bWire list the wire index of every point kept
// in pk.G1.A / pk.G1.B (the complement of pk.InfinityA / pk.InfinityB); kWire the private wires.
func (d *Device) LoadKey(pk *groth16_bn254.ProvingKey, aWire, bWire, kWire []uint32) (*C.zkmi_pk, error) {
	logN := 0
	for n := pk.Domain.Cardinality; n > 1; n >>= 1 {
		logN++
	}
	desc := C.zkmi_pk_desc{
		log_n: C.uint32_t(logN), n_wires: C.uint32_t(len(pk.InfinityA)),
		n_a: C.uint32_t(len(pk.G1.A)), n_b: C.uint32_t(len(pk.G1.B)),
		n_k: C.uint32_t(len(pk.G1.K)), n_z: C.uint32_t(len(pk.G1.Z)),
		a_wire: (*C.uint32_t)(unsafe.Pointer(&aWire[0])),
		b_wire: (*C.uint32_t)(unsafe.Pointer(&bWire[0])),
		k_wire: (*C.uint32_t)(unsafe.Pointer(&kWire[0])),
		g1_a:   unsafe.Pointer(&pk.G1.A[0]), g1_b: unsafe.Pointer(&pk.G1.B[0]),
		g1_k:   unsafe.Pointer(&pk.G1.K[0]), g1_z: unsafe.Pointer(&pk.G1.Z[0]),
		g2_b:   unsafe.Pointer(&pk.G2.B[0]),
		g1_alpha: unsafe.Pointer(&pk.G1.Alpha), g1_beta: unsafe.Pointer(&pk.G1.Beta),
		g1_delta: unsafe.Pointer(&pk.G1.Delta),
		g2_beta:  unsafe.Pointer(&pk.G2.Beta), g2_delta: unsafe.Pointer(&pk.G2.Delta),
	}
	d.mu.Lock()
	defer d.mu.Unlock()
	var h *C.zkmi_pk
	if rc := C.zkmi_pk_load(d.ctx, &desc, &h); rc != 0 {
		return nil, d.lastError()
	}
	return h, nil
}

// ProveBatch is groth16.Prove for `batch` independent witnesses of one circuit.  inputs holds the
// public then secret assignments of each witness back to back; rs the two blinding scalars gnark
// would sample inside Prove, per proof.  status[i] == -5: witness i does not satisfy the circuit.
func (d *Device) ProveBatch(pk *C.zkmi_pk, cs *C.zkmi_cs, inputs []fr.Element, batch int,
	rs []fr.Element, proofs []groth16_bn254.Proof) ([]int32, error) {
	raw := make([]byte, 256*batch) // Ar | Krs | Bs, the field order of gnark's Proof
	status := make([]int32, batch)
	d.mu.Lock()
	rc := C.zkmi_prove_batch(d.ctx, pk, cs, unsafe.Pointer(&inputs[0]), C.size_t(batch),
		unsafe.Pointer(&rs[0]), unsafe.Pointer(&raw[0]), (*C.int32_t)(unsafe.Pointer(&status[0])))
	d.mu.Unlock()
	if rc != 0 {
		return nil, d.lastError()
	}
	for i := range proofs {
		copy(unsafe.Slice((*byte)(unsafe.Pointer(&proofs[i].Ar)), 64), raw[256*i:])
		copy(unsafe.Slice((*byte)(unsafe.Pointer(&proofs[i].Krs)), 64), raw[256*i+64:])
		copy(unsafe.Slice((*byte)(unsafe.Pointer(&proofs[i].Bs)), 128), raw[256*i+128:])
	}
	return status, nil
}
