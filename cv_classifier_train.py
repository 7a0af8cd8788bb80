"""Image-only training entry point on the MI355X path: the loop of the reference's cv_classifier_train_daodian.py
(CvClassifier('efficientnet_b4', fc_dim=512) + ArcFace(m=0.2); torch.optim.Adam(lr=1e-3) over all parameters;
CosineAnnealingWarmRestarts(T_0=7, eta_min=1e-6) stepped per epoch; model.classifier.update_m(0.04) per epoch;
state-dict + optimiser checkpoints, :264-306).  BASELINE config 2 is this loop with EfficientNet-B0.

    python cv_classifier_train.py --synthetic --model efficientnet_b0 --num-labels 10000 --batch-size 128 --epochs 2 --steps-per-epoch 20

The reference's csv / jpg / albumentations input pipeline (:60-105, 198-241) is outside the hot path (OUT OF SCOPE);
synthetic batches of the same shape and dtype stand in for it here.
"""
import argparse
import os
import time

import torch
import torch.distributed as dist

from multimodalsimilar_amd import train as T


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="efficientnet_b4")                 # reference :190
    ap.add_argument("--fc-dim", type=int, default=512)                    # :50
    ap.add_argument("--no-fc", action="store_true")
    ap.add_argument("--num-labels", type=int, default=4181)               # :190
    ap.add_argument("--res", type=int, default=224)
    ap.add_argument("--batch-size", type=int, default=24)                 # :52
    ap.add_argument("--epochs", type=int, default=100)                    # :51
    ap.add_argument("--steps-per-epoch", type=int, default=100)
    ap.add_argument("--lr", type=float, default=1e-3)                     # :57
    ap.add_argument("--margin-step", type=float, default=0.04)            # :292
    ap.add_argument("--save-prefix", default=None, help="path prefix of the per-epoch state-dict / checkpoint files (:298-306)")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--log-every", type=int, default=10)
    args = ap.parse_args(argv)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("cv_classifier_train: needs an MI355X; the HIP path has no CPU fallback")
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)
    cfg = dict(kind="cv", image=args.model, res=args.res, batch=args.batch_size, classes=args.num_labels, fc_dim=args.fc_dim,
               use_fc=not args.no_fc)
    model = T.build_model(cfg, device, seed=0)
    loop = T.CvTrainLoop(model, lr=args.lr, margin_step=args.margin_step)
    step = 0
    for epoch in range(args.epochs):
        t0, avg = time.time(), 0.0
        for it in range(args.steps_per_epoch):
            batch = T.synthetic_batch(cfg, device, seed=1234 + rank + 1000003 * step)
            loss, pred = loop.step(batch)
            step += 1
            if step % args.log_every == 0:
                avg = float(loss.item())
                model.classifier.check_labels()
                if rank == 0:
                    print(f"epoch {epoch} step {step} loss {avg:.4f} lr {loop.optimizer.param_groups[0]['lr']:.3e} m {model.classifier.m:.2f}", flush=True)
        vloss, vpred = loop.evaluate(T.synthetic_batch(cfg, device, seed=99))
        loop.end_epoch()
        if rank == 0:
            dt = time.time() - t0
            print(f"epoch {epoch}: {world * args.batch_size * args.steps_per_epoch / dt:.1f} images/s, validation loss {float(vloss.item()):.4f}", flush=True)
            if args.save_prefix:
                torch.save(model.state_dict(), f"{args.save_prefix}_{epoch}.pt")                                             # :298
                torch.save({"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer": loop.optimizer.state_dict()},
                           f"{args.save_prefix}_{epoch}_checkpoints.pt")                                                     # :299-306
    if world > 1:
        dist.destroy_process_group()
    return model


if __name__ == "__main__":
    main()
