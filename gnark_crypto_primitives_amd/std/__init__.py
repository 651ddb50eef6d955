"""Restatements of the gnark std-lib gadgets the reference's packages import."""
