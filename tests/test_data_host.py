"""Host side of the data path (CPU): the reference's dataset / collate contract (multimodal_dataset.py:36-64,
multimodal_classifier_train.py:79-98) as restated in multimodalsimilar_amd.data -- decoded uint8 images + token tensors +
labels; the image transform itself runs on the GPU (tests/test_gpu_preprocess.py, tests/test_gpu_train_step.py)."""
import os

import numpy as np
import pytest
import torch

VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list("苹果香蕉牛奶面包新鲜进口大小红绿甜") + ["500", "g", "ml", "##g"]


def make_dataset(root, n=7):
    Image = pytest.importorskip("PIL.Image")
    import pandas as pd
    img_dir = os.path.join(root, "img")
    os.makedirs(img_dir, exist_ok=True)
    rng = np.random.default_rng(0)
    rows = []
    for i in range(n):
        h, w = int(rng.integers(40, 90)), int(rng.integers(40, 90))
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(img_dir, f"sku{i}.jpg"), quality=95)
        name = ("【福利秒杀】" if i % 2 else "") + "新鲜苹果[促销]500g" + "牛奶" * i
        rows.append(dict(spu_sn=f"sku{i}", spu_name=name, cateid=i % 3))
    csv = os.path.join(root, "train.csv")
    pd.DataFrame(rows).to_csv(csv, index=False)
    vocab = os.path.join(root, "vocab.txt")
    with open(vocab, "w", encoding="utf-8") as f:
        f.write("\n".join(VOCAB) + "\n")
    return csv, img_dir, vocab


def test_dataset_and_collate_follow_the_reference_contract(tmp_path):
    from multimodalsimilar_amd import data as D
    csv, img_dir, vocab = make_dataset(str(tmp_path))
    tok = D.load_tokenizer(vocab)
    assert tok.vocab_size == len(VOCAB)
    ds = D.MultimodalDataset(tokenizer=tok, transform=None, csv_path=csv, img_path=img_dir, use_label=True, max_length=32)
    assert len(ds) == 7
    img, t, label = ds[1]
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3
    assert len(t["input_ids"]) == 32 and t["input_ids"][0] == tok.cls_token_id and label.dtype == torch.int64
    # title cleaning (multimodal_dataset.py:21-31): the promo tag and the [..] span never reach the tokenizer
    assert D.preprocess_for_infer(["【福利秒杀】新鲜苹果[促销]500g"]) == ["新鲜苹果500g"]
    assert tok.decode(t["input_ids"], skip_special_tokens=True).replace(" ", "").startswith("新鲜苹果500g")
    loader = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False, collate_fn=D.collate_fn, num_workers=2, drop_last=True)
    batches = list(loader)
    assert len(batches) == 2
    b = batches[0]
    for k in ("input_ids", "token_type_ids", "attention_mask"):
        assert b[k].dtype == torch.int64 and tuple(b[k].shape) == (3, 32)
    assert b["labels"].tolist() == [0, 1, 2] and len(b["images"]) == 3 and all(i.dtype == torch.uint8 for i in b["images"])
    assert int(b["attention_mask"][0].sum()) < 32                    # padded to max_length, as the reference pads
    ds2 = D.MultimodalDataset(tokenizer=tok, transform=None, csv_path=csv, img_path=img_dir, use_label=False, max_length=32)
    assert "labels" not in D.collate_fn([ds2[0], ds2[1]])


def test_title_dataset_follows_the_text_only_contract(tmp_path):
    from multimodalsimilar_amd import data as D
    csv, _, vocab = make_dataset(str(tmp_path))
    tok = D.load_tokenizer(vocab)
    ds = D.TitleDataset(tok, csv, max_length=16)
    b = D.collate_titles([ds[0], ds[1], ds[4]])
    assert b["labels"].tolist() == [0, 1, 1] and all(tuple(b[k].shape) == (3, 16) and b[k].dtype == torch.int64
                                                     for k in ("input_ids", "token_type_ids", "attention_mask"))
    assert b["input_ids"][:, 0].tolist() == [tok.cls_token_id] * 3
