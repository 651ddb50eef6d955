"""MI355X-native Groth16/BN254 prover hot path behind the gadget API of
vocdoni/gnark-crypto-primitives."""
