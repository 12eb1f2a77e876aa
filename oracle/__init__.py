"""CPU oracle (test infrastructure only — see oracle/skani_oracle.h)."""
