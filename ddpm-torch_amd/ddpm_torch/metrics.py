"""``Evaluator`` placeholder (``ddpm_torch/metrics/*`` of tqch/ddpm-torch: FID and precision/recall on Inception-v3 features).

Evaluation needs the pretrained Inception network, torchvision and the dataset's reference statistics — none of which is
part of the accelerated hot path or available offline.  The name exists so that the reference's CLI
imports resolve; asking for an evaluation fails loudly instead of producing numbers from something else."""

__all__ = ["Evaluator"]


class Evaluator:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "FID / precision-recall evaluation is not part of the MI355X hot-path build (needs torchvision's Inception-v3 weights and "
            "dataset statistics); run training without --eval and evaluate the generated images with the reference's eval.py.")
