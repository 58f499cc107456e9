"""alac.net_amd -- MI355X-native ALAC frame-decode path (placeholder until the HIP library lands)."""
