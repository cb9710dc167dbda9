"""Shared tiny-model configs (must match tools/gen_fixtures.py)."""
VAE_CFG = dict(latent_channels=8, decoder_block_out_channels=(32, 64, 128), decoder_layers_per_block=(1, 1, 1, 2))
PIPE_DIT_CFG = dict(in_channels=8, out_channels=8, num_attention_heads=2, attention_head_dim=16, cross_attention_dim=32,
                    num_layers=3, caption_channels=32)
