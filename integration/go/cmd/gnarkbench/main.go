// gnarkbench times stock gnark (CPU) groth16.Prove on the north-star circuit, next to the same
// batch proved through libzkmi (SURVEY.md §8 f-2): the "reference CPU path timed on the same host"
// that BASELINE.json asks for.
//
// STATUS: SOURCE ONLY, UNBUILT AND UNTESTED (no Go toolchain and no module cache in the build
// image).  With Go >= 1.24 and the reference's go.mod (gnark v0.14.1-0.20251203003358-cce547909fed):
//
//	go run ./integration/go/cmd/gnarkbench -levels 160 -batch 64 -populated 10
//
// It builds the circuit of tree/test/verifier_bn254_test.go:23-34 with Siblings[levels], makes
// synthetic inclusion paths of the shape SURVEY.md §8d describes (k populated siblings, root
// folded with iden3 Poseidon as in tree/smt/verifier_level.go), runs groth16.Setup once, then
// (1) groth16.Prove per witness on all cores, (2) zkmi.Prover (Submit / Collect, two batches in
// flight) on GPU 0, verifies every proof
// with groth16.Verify, and prints both rates.
package main

import (
	"flag"
	"fmt"
	"math/big"
	"math/rand"
	"runtime"
	"sync"
	"time"

	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
	"github.com/consensys/gnark/backend/groth16"
	groth16_bn254 "github.com/consensys/gnark/backend/groth16/bn254"
	"github.com/consensys/gnark/backend/witness"
	cs_bn254 "github.com/consensys/gnark/constraint/bn254"
	"github.com/consensys/gnark/frontend"
	"github.com/consensys/gnark/frontend/cs/r1cs"
	"github.com/iden3/go-iden3-crypto/poseidon"

	"github.com/vocdoni/gnark-crypto-primitives/tree/smt"
	"github.com/vocdoni/gnark-crypto-primitives/utils"

	zkmi "github.com/vocdoni/gnark-crypto-primitives/integration/go"
)

type verifierCircuit struct {
	Root     frontend.Variable
	Key      frontend.Variable
	Value    frontend.Variable
	Siblings []frontend.Variable
}

func (c *verifierCircuit) Define(api frontend.API) error {
	valid := smt.InclusionVerifier(api, utils.PoseidonHasher, c.Root, c.Siblings, c.Key, c.Value)
	api.AssertIsEqual(valid, 1)
	return nil
}

// syntheticPath: `populated` random siblings at levels 0..populated-1, the rest zero; the root is
// folded from the leaf H(key, value, 1) upward, bit i of the key choosing the side at level i.
func syntheticPath(rng *rand.Rand, levels, populated int) (root, key, value *big.Int, sib []*big.Int) {
	key = new(big.Int).Rand(rng, new(big.Int).Lsh(big.NewInt(1), uint(levels)))
	value = new(big.Int).SetUint64(rng.Uint64())
	sib = make([]*big.Int, levels)
	for i := range sib {
		sib[i] = big.NewInt(0)
		if i < populated {
			sib[i] = new(big.Int).Rand(rng, fr.Modulus())
		}
	}
	cur, _ := poseidon.Hash([]*big.Int{key, value, big.NewInt(1)})
	for i := populated - 1; i >= 0; i-- {
		if key.Bit(i) == 1 {
			cur, _ = poseidon.Hash([]*big.Int{sib[i], cur})
		} else {
			cur, _ = poseidon.Hash([]*big.Int{cur, sib[i]})
		}
	}
	return cur, key, value, sib
}

func main() {
	levels := flag.Int("levels", 160, "tree depth")
	batch := flag.Int("batch", 64, "proofs")
	populated := flag.Int("populated", 10, "non-zero siblings per path")
	flag.Parse()

	circuit := &verifierCircuit{Siblings: make([]frontend.Variable, *levels)}
	ccs, err := frontend.Compile(ecc.BN254.ScalarField(), r1cs.NewBuilder, circuit)
	must(err)
	fmt.Printf("constraints: %d\n", ccs.GetNbConstraints())
	pk, vk, err := groth16.Setup(ccs)
	must(err)

	rng := rand.New(rand.NewSource(2))
	ws := make([]witness.Witness, *batch)
	for i := range ws {
		root, key, value, sib := syntheticPath(rng, *levels, *populated)
		a := &verifierCircuit{Root: root, Key: key, Value: value, Siblings: make([]frontend.Variable, *levels)}
		for j := range sib {
			a.Siblings[j] = sib[j]
		}
		ws[i], err = frontend.NewWitness(a, ecc.BN254.ScalarField())
		must(err)
	}

	// (1) stock gnark: one Prove per witness, as many at a time as there are cores
	t0 := time.Now()
	var wg sync.WaitGroup
	sem := make(chan struct{}, runtime.NumCPU())
	for i := range ws {
		wg.Add(1)
		sem <- struct{}{}
		go func(i int) {
			defer wg.Done()
			defer func() { <-sem }()
			proof, err := groth16.Prove(ccs, pk, ws[i])
			must(err)
			pub, _ := ws[i].Public()
			must(groth16.Verify(proof, vk, pub))
		}(i)
	}
	wg.Wait()
	cpu := time.Since(t0)
	fmt.Printf("gnark CPU: %d proofs in %v = %.2f proofs/s on %d cores\n", *batch, cpu,
		float64(*batch)/cpu.Seconds(), runtime.NumCPU())

	// (2) the same witnesses through libzkmi (gnark's solver + GPU quotient / MSM / assembly)
	dev, err := zkmi.Open(0)
	must(err)
	defer dev.Close()
	r1 := ccs.(*cs_bn254.R1CS)
	key, err := dev.LoadKey(pk.(*groth16_bn254.ProvingKey), r1, *batch, 0)
	must(err)
	mats, err := dev.LoadR1CS(r1) // L, R, O resident: a batch ships its wire vectors only
	must(err)
	prover, err := dev.NewProver(r1, key, mats, *batch)
	must(err)
	defer prover.Close()
	rs := make([]fr.Element, 2**batch)
	for i := range rs {
		rs[i].SetRandom()
	}
	// two batches in flight: batch k+1's CPU solve and PCIe transfer run under batch k's MSM kernels
	t0 = time.Now()
	must(prover.Submit(ws, rs))
	must(prover.Submit(ws, rs))
	proofs, status, err := prover.Collect()
	must(err)
	_, _, err = prover.Collect()
	must(err)
	gpu := time.Since(t0) / 2
	for _, st := range status {
		if st != 0 {
			panic("unsatisfied witness")
		}
	}
	for i := range proofs {
		pub, _ := ws[i].Public()
		must(groth16.Verify(&proofs[i], vk, pub))
	}
	fmt.Printf("zkmi GPU:  %d proofs in %v = %.2f proofs/s (all verified by gnark)\n", *batch, gpu,
		float64(*batch)/gpu.Seconds())
}

func must(err error) {
	if err != nil {
		panic(err)
	}
}
