"""Strong augmentations of the unlabeled branch with the reference's entry point
(u2pl/dataset/augmentation.py:471-541): rectangle / class draws stay on the host in the reference's order
(np.random for the boxes, torch.randperm for ClassMix), the mixing is one HIP launch."""
from ..trainer import classmix, classmix_select, cutmix, cutout, generate_cutmix_boxes


def generate_unsup_data(data, target, logits, mode="cutout"):
    B, _, im_h, im_w = data.shape
    target, logits = target.contiguous(), logits.contiguous()
    if mode == "classmix":
        return classmix(data, target, logits, classmix_select(target))
    boxes = generate_cutmix_boxes(B, im_h, im_w)
    if mode == "cutout":
        return cutout(data, target, logits, boxes)
    if mode == "cutmix":
        return cutmix(data, target, logits, boxes)
    raise ValueError(mode)
