#!/usr/bin/env python
"""Entry point: two-tower (image + text) ArcFace training on MI355X -- the drop-in for the reference's
``multimodal_classifier_train.py`` (loop :177-227, optimisers :152-164, eval :210-225, checkpoint :227).

The reference script does its heavy work at import time from hard-coded absolute paths and fetches a tokenizer by
name; here the same loop sits behind ``main()`` with flags whose defaults are the reference's constants
(batch 48, 30 epochs, 796 labels, lr 5e-5 / 1e-2, 15 % head warm-up, eval + whole-module checkpoint every 1000 steps).
Data: ``--train-csv/--img-dir`` need the reference's dataset stack (pandas, PIL, a local tokenizer vocab); without
them (or with ``--synthetic``) seeded synthetic batches of the same shapes/dtypes are used.
One process per GPU: launch with ``python -m torch.distributed.run --nproc-per-node N multimodal_classifier_train.py``
for data parallelism (RCCL all-reduce of the flat gradient buffers, overlapped with backward).
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from multimodalsimilar_amd import train as T  # noqa: E402

from multimodalsimilar_amd.data import MultimodalDataset, collate_fn, finish_batch, preprocess_for_infer, remove_words  # noqa: E402,F401


class RunningAccuracy:
    """torchmetrics.Accuracy stand-in with the reference's behaviour: a running mean that is never reset (E7);
    counters stay on the device, reading them is the only host sync."""

    def __init__(self, device):
        self.hit = torch.zeros((), dtype=torch.long, device=device)
        self.n = 0

    def __call__(self, pred, target):
        self.hit += (pred == target).sum()
        self.n += target.numel()

    def compute(self):
        return float(self.hit.item()) / max(1, self.n)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--batch-size", type=int, default=48)                 # reference :38
    ap.add_argument("--num-epochs", type=int, default=30)                 # reference :39
    ap.add_argument("--steps-per-epoch", type=int, default=100, help="synthetic data: steps per epoch")
    ap.add_argument("--num-labels", type=int, default=796)                # reference :127
    ap.add_argument("--text-model", default="large", choices=["base", "large", "tiny"])
    ap.add_argument("--image-model", default="efficientnet_b4")
    ap.add_argument("--res", type=int, default=224)
    ap.add_argument("--seq-len", type=int, default=128)                   # multimodal_dataset.py:47
    ap.add_argument("--use-fc", action="store_true", help="CvClassifier top: Dropout -> Linear(fc_dim) -> BatchNorm1d")
    ap.add_argument("--fc-dim", type=int, default=512)
    ap.add_argument("--cv-classifier-path", default=None, help="whole-module pickle of a CvClassifier (reference :124)")
    ap.add_argument("--nlp-classifier-path", default=None, help="whole-module pickle of an NlpClassifier (reference :125)")
    ap.add_argument("--eval-every", type=int, default=1000)               # reference :210
    ap.add_argument("--eval-batches", type=int, default=4)
    ap.add_argument("--save-dir", default=None, help="directory for {step}.pt whole-module checkpoints (reference :227)")
    ap.add_argument("--literal-loss", action="store_true", help="materialised logits + nn.CrossEntropyLoss (the reference's literal path)")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--train-csv", default=None, help="csv with spu_sn, spu_name, cateid columns (reference :128-133)")
    ap.add_argument("--test-csv", default=None)
    ap.add_argument("--img-dir", default=None, help="directory holding {spu_sn}.jpg")
    ap.add_argument("--vocab", default=None, help="local BERT vocab.txt (the reference fetches hfl/chinese-roberta-wwm-ext by name)")
    ap.add_argument("--num-workers", type=int, default=16)                # reference :141
    ap.add_argument("--log-every", type=int, default=10)
    ap.add_argument("--max-steps", type=int, default=None)
    args = ap.parse_args(argv)

    # before the first HIP call of the process: the host driver only supports dmabuf IPC (RCCL / cross-process tensors)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        raise SystemExit("multimodal_classifier_train: needs an MI355X; the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)

    cfg = dict(kind="multimodal", text=args.text_model, image=args.image_model, res=args.res, seq_len=args.seq_len,
               batch=args.batch_size, classes=args.num_labels, use_fc=args.use_fc, fc_dim=args.fc_dim)
    if args.cv_classifier_path and args.nlp_classifier_path:
        from multimodal_classifier import MultimodalClassifier
        cv = torch.load(args.cv_classifier_path, weights_only=False)
        nlp = torch.load(args.nlp_classifier_path, weights_only=False)
        emb = (cv.fc.out_features if cv.use_fc else cv.backbone.num_features) + nlp.ptm.config.hidden_size
        model = MultimodalClassifier(device, cv, nlp, emb_size=emb, num_labels=args.num_labels)
    else:
        model = T.build_model(cfg, device, seed=0)

    loaders = None
    if args.train_csv and not args.synthetic:
        if not (args.img_dir and args.vocab):
            raise SystemExit("--train-csv needs --img-dir and --vocab (a local vocab.txt; nothing can be downloaded)")
        from torch.utils.data import DataLoader, DistributedSampler
        from multimodalsimilar_amd.data import load_tokenizer
        from multimodalsimilar_amd.preprocess import create_transform
        tokenizer = load_tokenizer(args.vocab)
        transform_eff = create_transform(input_size=(3, args.res, args.res), interpolation="bicubic", crop_pct=1.0, device=device)
        def make(csv, shuffle):
            ds = MultimodalDataset(tokenizer=tokenizer, transform=transform_eff, csv_path=csv, img_path=args.img_dir,
                                   use_label=True, max_length=args.seq_len)
            sampler = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=shuffle) if world > 1 else None
            return DataLoader(ds, shuffle=shuffle and sampler is None, sampler=sampler, batch_size=args.batch_size,
                              collate_fn=collate_fn, num_workers=args.num_workers, drop_last=True)
        loaders = (make(args.train_csv, True), make(args.test_csv, False) if args.test_csv else None)
        args.steps_per_epoch = len(loaders[0])
        if tokenizer.vocab_size > model.nlp.ptm.config.vocab_size:
            raise SystemExit(f"vocab.txt has {tokenizer.vocab_size} entries, the text tower's embedding table {model.nlp.ptm.config.vocab_size}")
    num_training_steps = args.num_epochs * args.steps_per_epoch          # reference :150
    step = T.TrainStep(model, "multimodal", num_training_steps, fused_loss=not args.literal_loss)
    train_accuracy, test_accuracy = RunningAccuracy(device), RunningAccuracy(device)
    writer = None
    try:
        from torch.utils.tensorboard import SummaryWriter                # reference :121 (optional here)
        writer = SummaryWriter() if rank == 0 else None
    except Exception:
        pass

    global_step, test_acc, t_last = 0, 0.0, time.time()
    for epoch in range(args.num_epochs):
        if loaders and world > 1:
            loaders[0].sampler.set_epoch(epoch)       # before the iterator draws its indices, or the epoch's shuffle lags by one
        data_iter = iter(loaders[0]) if loaders else None
        for it in range(args.steps_per_epoch):
            if data_iter is not None:
                batch = finish_batch(next(data_iter), transform_eff, device)
            else:
                batch = T.synthetic_batch(cfg, device, seed=1234 + rank + 1000003 * global_step)
            loss, pred = step.step(batch)                                 # :179-201
            train_accuracy(pred, batch["labels"])                        # :191-192
            global_step += 1
            if global_step % args.log_every == 0:                        # :203-208, without the per-step host syncs
                lv, acc = float(loss.item()), train_accuracy.compute()
                model.classifier.check_labels()
                if rank == 0:
                    dt = (time.time() - t_last) / args.log_every
                    print(f"step {global_step} loss {lv:.4f} acc {acc:.4f} test_acc {test_acc:.4f} "
                          f"{world * args.batch_size / dt:.1f} pairs/s", flush=True)
                    if writer:
                        writer.add_scalar('Loss/train', lv, global_step)
                        writer.add_scalar('Acc/train', acc, global_step)
                t_last = time.time()
            if global_step % args.eval_every == 0:                        # :210-225
                model.eval()
                test_iter = iter(loaders[1]) if loaders and loaders[1] is not None else None
                for eb in range(args.eval_batches if test_iter is None else min(args.eval_batches, len(loaders[1]))):
                    tb = finish_batch(next(test_iter), transform_eff, device) if test_iter is not None else T.synthetic_batch(cfg, device, seed=99 + eb)
                    with torch.no_grad():
                        cos = model(**{**T.model_inputs("multimodal", tb), "is_test": True})
                    test_accuracy(torch.argmax(cos, dim=-1), tb["labels"])
                test_acc = test_accuracy.compute()
                if writer:
                    writer.add_scalar('Acc/test', test_acc, global_step)
                if args.save_dir and rank == 0:
                    os.makedirs(args.save_dir, exist_ok=True)
                    torch.save(model, os.path.join(args.save_dir, f"{global_step}.pt"))   # :227
            if args.max_steps and global_step >= args.max_steps:
                break
        if args.max_steps and global_step >= args.max_steps:
            break
    if world > 1:
        dist.destroy_process_group()
    return model


if __name__ == "__main__":
    main()
