"""Two training steps of the image tower alone (EfficientNet-B4 @ 224, B = 256, ArcFace 1000 classes) for rocprofv3 --pmc passes."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
warnings.simplefilter("ignore")
from cv_classifier import CvClassifier
torch.manual_seed(0)
m = CvClassifier("efficientnet_b4", 512, 1000, pretrained=False, use_fc=False).to("cuda").train()
x = torch.randn(256, 3, 224, 224, device="cuda")
y = torch.randint(0, 1000, (256,), device="cuda")
for _ in range(int(os.environ.get("PMC_STEPS", "2"))):
    loss, _ = m.forward_loss(x, y)
    loss.backward()
torch.cuda.synchronize()
print("loss", float(loss))
