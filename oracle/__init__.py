"""CPU oracle — test infrastructure only. See oracle/oracle.cpp."""
