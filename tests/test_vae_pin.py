"""Pin of the MuseTalk VAE restatement (oracle/musetalk_oracle.py vae_decode / vae_encode_moments) against an INDEPENDENT
third-party implementation of the same network that IS installed here: transformers' Janus VQ-VAE encoder / decoder
(transformers/models/janus/modeling_janus.py), i.e. the taming-transformers / latent-diffusion autoencoder that diffusers'
AutoencoderKL is a port of: conv_in, mid block (ResnetBlock, single-head AttnBlock with 1x1 q/k/v/proj and C^-0.5 scale,
ResnetBlock), per level num_res_blocks(+1) ResnetBlocks (GroupNorm 32 eps 1e-6, swish, 1x1 nin_shortcut when channels
change), nearest-2x + 3x3 conv upsample / (0,1,0,1)-pad stride-2 downsample, norm_out, swish, conv_out.

The same seeded weights are loaded into both: into the oracle under diffusers' AutoencoderKL key names (what a real
sd-vae-ft-mse checkpoint uses, synth_inputs.py), into the Janus modules under their own names through the explicit
name map below.  Equal outputs pin the oracle's WIRING (block order, channel plan, resnet / attention / resample
composition, eps, activation) to code nobody here wrote.  What this cannot pin: diffusers' key names and config values
themselves (sd-vae-ft-mse config: block_out_channels (128,256,512,512), layers_per_block 2, latent 4, no attention outside
the mid block) - stated in oracle/musetalk_oracle.py.  The reference calls AutoencoderKL at
avatars/musetalk/models/vae.py:24,92,103.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
janus = pytest.importorskip("transformers.models.janus.modeling_janus")

from oracle import musetalk_oracle as M  # noqa: E402
import synth_inputs as synth


def _cfg():
    from transformers.models.janus.configuration_janus import JanusVQVAEConfig
    return JanusVQVAEConfig(embed_dim=4, num_embeddings=16, double_latent=True, latent_channels=4, in_channels=3, out_channels=3,
                            base_channels=128, channel_multiplier=[1, 2, 4, 4], num_res_blocks=2, dropout=0.0)


def _load_resnet(mod, sd, p):
    mod.norm1.weight.data, mod.norm1.bias.data = sd[p + ".norm1.weight"], sd[p + ".norm1.bias"]
    mod.norm2.weight.data, mod.norm2.bias.data = sd[p + ".norm2.weight"], sd[p + ".norm2.bias"]
    mod.conv1.weight.data, mod.conv1.bias.data = sd[p + ".conv1.weight"], sd[p + ".conv1.bias"]
    mod.conv2.weight.data, mod.conv2.bias.data = sd[p + ".conv2.weight"], sd[p + ".conv2.bias"]
    if hasattr(mod, "nin_shortcut"):
        mod.nin_shortcut.weight.data, mod.nin_shortcut.bias.data = sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"]
    else:
        assert p + ".conv_shortcut.weight" not in sd


def _load_attn(mod, sd, p):
    mod.norm.weight.data, mod.norm.bias.data = sd[p + ".group_norm.weight"], sd[p + ".group_norm.bias"]
    for theirs, ours in (("q", "to_q"), ("k", "to_k"), ("v", "to_v"), ("proj_out", "to_out.0")):
        conv = getattr(mod, theirs)
        conv.weight.data = sd[f"{p}.{ours}.weight"].reshape(conv.weight.shape)      # Linear (C,C) -> 1x1 conv (C,C,1,1)
        conv.bias.data = sd[f"{p}.{ours}.bias"]


def _conv(mod, sd, p):
    mod.weight.data, mod.bias.data = sd[p + ".weight"], sd[p + ".bias"]


def test_vae_decoder_restatement_vs_transformers_vqvae_decoder():
    sd = {k: torch.from_numpy(v) for k, v in synth.vae_decoder_state_dict().items()}
    dec = janus.JanusVQVAEDecoder(_cfg()).eval()
    for lvl in dec.up:                                    # sd-vae has attention in the mid block only
        lvl.attn = torch.nn.ModuleList()
    _conv(dec.conv_in, sd, "decoder.conv_in")
    _load_resnet(dec.mid.block_1, sd, "decoder.mid_block.resnets.0")
    _load_attn(dec.mid.attn_1, sd, "decoder.mid_block.attentions.0")
    _load_resnet(dec.mid.block_2, sd, "decoder.mid_block.resnets.1")
    for i in range(4):                                    # Janus up[i] runs i-th, like diffusers up_blocks[i]
        for j in range(3):
            _load_resnet(dec.up[i].block[j], sd, f"decoder.up_blocks.{i}.resnets.{j}")
        if i < 3:
            _conv(dec.up[i].upsample.conv, sd, f"decoder.up_blocks.{i}.upsamplers.0.conv")
    _conv(dec.conv_norm_out if hasattr(dec, "conv_norm_out") else dec.norm_out, sd, "decoder.conv_norm_out")
    _conv(dec.conv_out, sd, "decoder.conv_out")
    g = torch.Generator().manual_seed(3)
    z = torch.randn(2, 4, 8, 8, generator=g) * 3.0         # the graph is resolution-agnostic: 8x8 latents -> 64x64 images
    with torch.no_grad():
        post = torch.nn.functional.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
        theirs = dec(post.clone())
        taps = {}
        ours = M.vae_decode(sd, z, taps)
    assert ours.shape == theirs.shape == (2, 3, 64, 64)
    err = float((ours - theirs).abs().max()) / float(theirs.abs().max())
    print(f"[vae pin] decoder vs transformers JanusVQVAEDecoder: rel max err {err:.2e}")
    assert err < 1e-5


def test_vae_encoder_restatement_vs_transformers_vqvae_encoder():
    if not hasattr(synth, "vae_encoder_state_dict"):
        pytest.skip("no synthetic encoder weights")
    sd = {k: torch.from_numpy(v) for k, v in synth.vae_encoder_state_dict().items()}
    enc = janus.JanusVQVAEEncoder(_cfg()).eval()
    for lvl in enc.down:
        lvl.attn = torch.nn.ModuleList()
    _conv(enc.conv_in, sd, "encoder.conv_in")
    for i in range(4):
        for j in range(2):
            _load_resnet(enc.down[i].block[j], sd, f"encoder.down_blocks.{i}.resnets.{j}")
        if i < 3:
            _conv(enc.down[i].downsample.conv, sd, f"encoder.down_blocks.{i}.downsamplers.0.conv")
    _load_resnet(enc.mid.block_1, sd, "encoder.mid_block.resnets.0")
    _load_attn(enc.mid.attn_1, sd, "encoder.mid_block.attentions.0")
    _load_resnet(enc.mid.block_2, sd, "encoder.mid_block.resnets.1")
    _conv(enc.norm_out, sd, "encoder.conv_norm_out")
    _conv(enc.conv_out, sd, "encoder.conv_out")
    g = torch.Generator().manual_seed(4)
    x = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    with torch.no_grad():
        theirs = torch.nn.functional.conv2d(enc(x.clone()), sd["quant_conv.weight"], sd["quant_conv.bias"])
        ours = M.vae_encode_moments(sd, x)
    assert ours.shape == theirs.shape == (2, 8, 8, 8)
    err = float((ours - theirs).abs().max()) / float(theirs.abs().max())
    print(f"[vae pin] encoder vs transformers JanusVQVAEEncoder: rel max err {err:.2e}")
    assert err < 1e-5


def test_deprecated_attention_keys_are_converted():
    """sd-vae-ft-mse stores the mid-block attention as query / key / value / proj_attn: the plugin renames them
    the way AutoencoderKL.from_pretrained does before the engine looks tensors up by the current names."""
    from livetalking_amd.avatars.musetalk_avatar import convert_deprecated_vae_attention
    new = synth.vae_decoder_state_dict()
    old = {}
    for k, v in new.items():
        k2 = k
        for o, n in (("query", "to_q"), ("key", "to_k"), ("value", "to_v"), ("proj_attn", "to_out.0")):
            if f".attentions.0.{n}." in k:
                k2 = k.replace(f".{n}.", f".{o}.")
                if k.endswith(".weight") and o != "proj_attn":
                    v = v.reshape(v.shape[0], v.shape[1], 1, 1)        # some exports keep 1x1-conv shaped projections
        old[k2] = v
    assert any(".query." in k for k in old) and not any(".to_q." in k for k in old)
    back = convert_deprecated_vae_attention(old)
    assert sorted(back) == sorted(new)
    for k in new:
        assert back[k].shape == new[k].shape and np.array_equal(np.asarray(back[k]), new[k]), k
