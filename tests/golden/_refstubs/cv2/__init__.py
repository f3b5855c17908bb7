"""Import stub (generator-only)."""
