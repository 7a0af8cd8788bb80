#!/usr/bin/env python
"""Entry point: text-only ArcFace training -- drop-in for the reference's ``nlp_classifier_train.py``
(loop :110-159; optimisers :89-97: AdamW(emb_layer, 5e-5) linear/no warm-up + AdamW(classifier, 1e-2) linear/15 %
warm-up; batch 256; eval every 100 steps; whole-module checkpoint every 1000).  BASELINE.json config 1 names it
(roberta-base, seq_len 64, bs 8, 1k classes).  ``--train-csv/--vocab`` read the reference's csv format (spu_name, cateid) through
DataLoader workers; without them seeded synthetic token batches of the same shapes are used."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from multimodalsimilar_amd import train as T  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--batch-size", type=int, default=256)               # reference :32
    ap.add_argument("--num-epochs", type=int, default=10)
    ap.add_argument("--steps-per-epoch", type=int, default=100)
    ap.add_argument("--num-labels", type=int, default=796)               # reference :65
    ap.add_argument("--text-model", default="base", choices=["base", "large", "tiny"])
    ap.add_argument("--seq-len", type=int, default=128)
    ap.add_argument("--eval-every", type=int, default=100)               # reference :142
    ap.add_argument("--save-dir", default=None)
    ap.add_argument("--max-steps", type=int, default=None)
    ap.add_argument("--log-every", type=int, default=10)
    ap.add_argument("--train-csv", default=None, help="csv with spu_name, cateid columns (reference :78-87)")
    ap.add_argument("--vocab", default=None, help="local BERT vocab.txt (the reference fetches hfl/chinese-roberta-wwm-ext by name)")
    ap.add_argument("--num-workers", type=int, default=4)
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("nlp_classifier_train: needs an MI355X; the HIP path has no CPU fallback")
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(device)
    cfg = dict(kind="nlp", text=args.text_model, seq_len=args.seq_len, batch=args.batch_size, classes=args.num_labels)
    model = T.build_model(cfg, device, seed=0)
    loader = None
    if args.train_csv:
        if not args.vocab:
            raise SystemExit("--train-csv needs --vocab (a local vocab.txt; nothing can be downloaded)")
        from torch.utils.data import DataLoader
        from multimodalsimilar_amd.data import TitleDataset, collate_titles, load_tokenizer
        tokenizer = load_tokenizer(args.vocab)
        if tokenizer.vocab_size > model.ptm.config.vocab_size:
            raise SystemExit(f"vocab.txt has {tokenizer.vocab_size} entries, the text tower's embedding table {model.ptm.config.vocab_size}")
        loader = DataLoader(TitleDataset(tokenizer, args.train_csv, max_length=args.seq_len), shuffle=True, batch_size=args.batch_size,
                            collate_fn=collate_titles, num_workers=args.num_workers, drop_last=True)
        args.steps_per_epoch = len(loader)
    step = T.TrainStep(model, "nlp", args.num_epochs * args.steps_per_epoch)
    hit = torch.zeros((), dtype=torch.long, device=device)
    n, gs, t0 = 0, 0, time.time()
    for epoch in range(args.num_epochs):
        data_iter = iter(loader) if loader is not None else None
        for it in range(args.steps_per_epoch):
            if data_iter is not None:
                batch = {k: v.to(device, non_blocking=True) for k, v in next(data_iter).items()}
            else:
                batch = T.synthetic_batch(cfg, device, seed=1234 + 1000003 * gs)
            loss, pred = step.step(batch)
            hit += (pred == batch["labels"]).sum(); n += pred.numel(); gs += 1
            if gs % args.log_every == 0:
                print(f"step {gs} loss {float(loss.item()):.4f} acc {float(hit.item()) / n:.4f} "
                      f"{args.batch_size * args.log_every / (time.time() - t0):.1f} titles/s", flush=True)
                t0 = time.time()
            if gs % args.eval_every == 0:
                model.eval()
                with torch.no_grad():
                    tb = T.synthetic_batch(cfg, device, seed=7)
                    model(**{**T.model_inputs("nlp", tb), "is_test": True})
            if args.save_dir and gs % 1000 == 0:
                os.makedirs(args.save_dir, exist_ok=True)
                torch.save(model, os.path.join(args.save_dir, f"{gs}.pt"))
            if args.max_steps and gs >= args.max_steps:
                return model
    return model


if __name__ == "__main__":
    main()
