"""MI355X-native rigid-body-dynamics code generator (gfx950 / CDNA4, HIP, wave64)."""
