"""Dev tool (GPU): which trainable parameters of the training-step models never receive a gradient?  DDP needs
find_unused_parameters=True for exactly those models (HD_Xray_Pretrain_MAE/pretrain/main.py:183); the models carry the answer as
`ddp_find_unused_parameters`, tests/test_ddp_finetune_gpu.py holds them to it.  usage: python tools/unused_params.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

DEV = "cuda:0"


def report(name, model, loss):
    loss.backward()
    idle = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
    n_train = sum(1 for p in model.parameters() if p.requires_grad)
    flag = any(getattr(m, "ddp_find_unused_parameters", False) for m in model.modules())
    print(f"{name}: {n_train} trainable tensors, {len(idle)} without a gradient, flag={flag}: {idle[:12]}")


def main():
    from test_mambaxray_vl import WordTokenizer, _samples
    from medical_image_analysis_amd import mambaxray_vl as mx
    from medical_image_analysis_amd.r2gencsr import R2GenCSR
    from medical_image_analysis_amd.vmamba import VSSM, vssm1_base_0229
    from medical_image_analysis_amd.models_pretrain import VisionMamba
    from medical_image_analysis_amd.models_mamba import arm_base_pz16
    from medical_image_analysis_amd.mae import MaskedAutoencoderViT
    tiny = dict(vocab_size=256, hidden_size=128, intermediate_size=352, num_hidden_layers=2, num_attention_heads=2,
                num_key_value_heads=2, max_position_embeddings=1024)
    torch.manual_seed(0)
    m = VisionMamba(img_size=128, patch_size=16, stride=16, embed_dim=128, depth=12, dec_embed_dim=128, rms_norm=True,
                    residual_in_fp32=True, fused_add_norm=True, if_abs_pos_embed=True, bimamba_type="None").to(DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        report("VisionMamba", m, m(torch.randn(2, 3, 128, 128, device=DEV)).mean())
    m = arm_base_pz16("base", drop_path_rate=0.0).to(DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        report("ARM-base v3", m, m(torch.randn(2, 3, 224, 224, device=DEV)).float().square().mean())
    m = MaskedAutoencoderViT(img_size=64, patch_size=16, in_chans=1, embed_dim=64, depth=2, num_heads=4, decoder_embed_dim=64,
                             decoder_depth=1, decoder_num_heads=4).to(DEV)
    with torch.autocast("cuda", dtype=torch.float16):
        loss, mask = m(torch.randn(2, 1, 64, 64, device=DEV), 0, 0.75, 0.0)
        report("MAE", m, (loss * mask).sum() / mask.sum())
    m = vssm1_base_0229(drop_path_rate=0.0).to(DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        report("VSSM global_features", m, m(torch.randn(2, 3, 224, 224, device=DEV), global_features=True).float().square().mean())
    for freeze in (False, True):
        llm = mx.build_report_decoder(tiny)
        args = mx.default_args(vision_model="Base-None", max_length=16, freeze_vm=freeze)
        m = mx.MambaXrayVLDownStream(args, tokenizer=WordTokenizer(), llm=llm).to(DEV)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            report(f"MambaXrayVLDownStream freeze_vm={freeze}", m, m(_samples(2))["loss"])
    enc = VSSM(depths=[1, 1, 2, 1], dims=32, ssm_d_state=1, ssm_ratio=2.0, ssm_conv=3, ssm_conv_bias=False, forward_type="v3noz",
               mlp_ratio=4.0, downsample_version="v3", patchembed_version="v2", drop_path_rate=0.0)
    llm = mx.build_report_decoder(tiny, dtype=torch.bfloat16)
    args = mx.default_args(max_length=16, context_pair=3, freeze_vm=False, llm_freeze=True,
                           positive="Note: <Img><ImageHere></Img> with disease .", negative="Note: <Img><ImageHere></Img> is healthy .",
                           use_feature_mean=True, instruction="Generate a report .")
    m = R2GenCSR(args, tokenizer=WordTokenizer(), llm=llm, encoder=enc).to(DEV)
    m.set_context_samples(torch.randn(3, 3, 224, 224).to(DEV), torch.randn(3, 3, 224, 224).to(DEV))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        report("R2GenCSR linear", m, m(_samples(2))["loss"])
    args = mx.default_args(max_length=16, context_pair=0, freeze_vm=False, llm_freeze=True, proj="qformer", instruction="Generate a report .")
    m = R2GenCSR(args, tokenizer=WordTokenizer(), llm=mx.build_report_decoder(tiny, dtype=torch.bfloat16), encoder=VSSM(
        depths=[1, 1, 2, 1], dims=32, ssm_d_state=1, ssm_ratio=2.0, ssm_conv=3, ssm_conv_bias=False, forward_type="v3noz",
        mlp_ratio=4.0, downsample_version="v3", patchembed_version="v2", drop_path_rate=0.0)).to(DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        report("R2GenCSR qformer", m, m(_samples(2))["loss"])


if __name__ == "__main__":
    main()
