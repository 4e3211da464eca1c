"""CPU oracle for the BALM 2.0 hot path -- TEST INFRASTRUCTURE ONLY (see balm_oracle.hpp)."""
