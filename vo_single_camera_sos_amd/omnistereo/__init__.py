"""Host-side mirror of the reference's `omnistereo` package for the VO hot path only (same module, class
and method names for that path; everything else of the reference package is out of scope)."""
