"""The batch contract of the reference's dataset + collate, with the image transform moved to the GPU.

Reference: ``MultimodalDataset`` (multimodal_dataset.py:36-64: one csv row = spu_sn, spu_name, cateid; ``{img_path}/{spu_sn}.jpg``
decoded with PIL, title cleaned by ``preprocess_for_infer`` and tokenised with padding='max_length', max_length=128,
truncation=True) and ``collate_fn`` (multimodal_classifier_train.py:79-98: DataCollatorWithPadding over the token dicts +
``img_tensor`` = stack of transformed images + ``labels``), 16 DataLoader workers running the timm transform on the CPU.

Here the workers only decode (PIL) and tokenise; the collated batch carries the decoded uint8 images and
``finish_batch`` runs resize / crop / normalise for the whole batch on the GPU (multimodalsimilar_amd.preprocess) and moves
the token tensors over.  The resulting dict has the reference's keys and dtypes: ``input_ids``, ``token_type_ids``,
``attention_mask`` int64 [B, 128], ``img_tensor`` fp32 [B, 3, S, S], ``labels`` int64 [B].
"""
import re

import numpy as np
import torch

remove_words = ['【福利秒杀】', '【每日福利】', '【福利爆款】', '【专柜品质】', '【1元秒杀】', '【直播专用1元秒杀】', '【', '】', '源本']


def preprocess_for_infer(spu_names):
    """Title cleaning of the reference (multimodal_dataset.py:21-31): drop promo tags and [...] spans."""
    out = []
    for line in spu_names:
        for r in remove_words:
            line = line.replace(r, '')
        for c in re.findall(r'\[[^()]*\]', line):
            line = line.replace(c, '')
        out.append(line)
    return out


def load_tokenizer(vocab_path):
    """BertTokenizer over a LOCAL vocab.txt (the reference fetches 'hfl/chinese-roberta-wwm-ext' by name,
    multimodal_classifier_train.py:76; nothing can be downloaded here).  transformers 5 takes the vocabulary itself, 4 a path."""
    from transformers import BertTokenizer
    with open(vocab_path, encoding="utf-8") as f:
        vocab = {tok.rstrip("\n"): i for i, tok in enumerate(f) if tok.rstrip("\n")}
    try:
        tk = BertTokenizer(vocab=vocab)
    except TypeError:
        tk = BertTokenizer(vocab_file=vocab_path)
    if tk.vocab_size != len(vocab):
        raise ValueError(f"tokenizer built from {vocab_path} has {tk.vocab_size} entries, the file {len(vocab)}")
    return tk


class MultimodalDataset(torch.utils.data.Dataset):
    """Same constructor as the reference's class; ``transform`` is kept for signature compatibility but applied later, on the
    GPU, by ``finish_batch`` (pass the ``create_transform(...)`` object there).  Items: (uint8 [H, W, 3] image, token dict[, label])."""

    def __init__(self, tokenizer, transform, csv_path, img_path, use_label=False, max_length=128):
        from pandas import read_csv
        self.dataframe = read_csv(csv_path)
        self.csv_path, self.img_path = csv_path, img_path
        self.tokenizer, self.transform, self.use_label, self.max_length = tokenizer, transform, use_label, max_length

    def tokenize_function(self, spu_name):
        return self.tokenizer(text=preprocess_for_infer([spu_name])[0], padding="max_length", max_length=self.max_length,
                              truncation=True)

    def __getitem__(self, index):
        from PIL import Image
        spusn = self.dataframe['spu_sn'][index]
        img = np.array(Image.open("{}/{}.jpg".format(self.img_path, spusn)).convert('RGB'))
        tok = self.tokenize_function(self.dataframe['spu_name'][index])
        if self.use_label:
            return img, tok, torch.tensor(int(self.dataframe['cateid'][index]), dtype=torch.int64)
        return img, tok

    def __len__(self):
        return len(self.dataframe)


def collate_fn(batch):
    """Token dicts -> int64 tensors (already padded to max_length, so DataCollatorWithPadding's job reduces to stacking),
    images stay a list of uint8 arrays (ragged sizes), labels stacked when present."""
    out = {}
    toks = [b[1] for b in batch]
    for k in ("input_ids", "token_type_ids", "attention_mask"):
        out[k] = torch.tensor([list(t[k]) for t in toks], dtype=torch.int64)
    out["images"] = [torch.from_numpy(b[0]) for b in batch]
    if len(batch[0]) > 2:
        out["labels"] = torch.stack([b[2] for b in batch])
    return out


def finish_batch(collated, transform, device):
    """Host batch -> the reference's batch dict on the GPU; ``transform`` = multimodalsimilar_amd.preprocess.create_transform(...)."""
    out = {k: v.to(device, non_blocking=True) for k, v in collated.items() if k != "images"}
    out["img_tensor"] = transform.batch(collated["images"])
    return out


class TitleDataset(torch.utils.data.Dataset):
    """Text-only rows of the reference's nlp_classifier_train.py (:78-87: csv with ``spu_name`` and ``cateid``; titles cleaned by
    ``preprocess_for_infer``, tokenised with padding='max_length', max_length=128, truncation=True; ``cateid`` becomes ``labels``)."""

    def __init__(self, tokenizer, csv_path, text_col="spu_name", label_col="cateid", max_length=128):
        from pandas import read_csv
        self.dataframe = read_csv(csv_path)
        self.tokenizer, self.text_col, self.label_col, self.max_length = tokenizer, text_col, label_col, max_length

    def __len__(self):
        return len(self.dataframe)

    def __getitem__(self, index):
        tok = self.tokenizer(text=preprocess_for_infer([str(self.dataframe[self.text_col][index])])[0], padding="max_length",
                             max_length=self.max_length, truncation=True)
        return tok, torch.tensor(int(self.dataframe[self.label_col][index]), dtype=torch.int64)


def collate_titles(batch):
    out = {k: torch.tensor([list(t[k]) for t, _ in batch], dtype=torch.int64) for k in ("input_ids", "token_type_ids", "attention_mask")}
    out["labels"] = torch.stack([y for _, y in batch])
    return out
