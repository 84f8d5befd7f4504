"""studies that run the oracle (test infrastructure): the CPU rounding-policy simulator and attention diagnostics; not collected by pytest."""
