"""lbm_amd -- MI355X-native latent retrieval."""
