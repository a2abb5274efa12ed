"""Ray / sphere / volume geometry of the hot path, kernel-backed."""
