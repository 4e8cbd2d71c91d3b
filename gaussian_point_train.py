"""Train a Gaussian point-cloud scene on MI355X -- same command line as the reference's
``gaussian_point_train.py`` (``--train_config cfg.yaml`` / ``--gen_template_only``).

Single GPU:  python gaussian_point_train.py --train_config config/my_scene.yaml
Multi GPU :  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
                 gaussian_point_train.py --train_config config/my_scene.yaml
             (tile rows of every frame are sharded over the ranks; see taichi_3d_gaussian_splatting_amd/distributed.py)
"""
import argparse
import logging
import os

import torch

from taichi_3d_gaussian_splatting_amd.GaussianPointTrainer import GaussianPointCloudTrainer


def main() -> None:
    parser = argparse.ArgumentParser("Train a Gaussian Point Cloud Scene")
    parser.add_argument("--train_config", type=str, required=True)
    parser.add_argument("--gen_template_only", action="store_true", default=False)
    args = parser.parse_args()
    if args.gen_template_only:
        GaussianPointCloudTrainer.TrainConfig().to_yaml_file(args.train_config)
        return
    logging.basicConfig(level=logging.INFO)
    config = GaussianPointCloudTrainer.TrainConfig.from_yaml_file(args.train_config)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        torch.distributed.init_process_group("nccl")   # RCCL on ROCm
    if torch.cuda.is_available():   # the launching threads on one L3 complex next to the GPU (host_affinity.py)
        from taichi_3d_gaussian_splatting_amd import host_affinity
        host_affinity.pin_host_threads(torch.cuda.current_device())
    GaussianPointCloudTrainer(config).train()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
