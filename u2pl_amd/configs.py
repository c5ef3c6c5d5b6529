"""Built-in copies of the reference experiment settings used by bench / smoke
(values from experiments/cityscapes/744/ours/config.yaml and
experiments/pascal/1464/ours/config.yaml; any reference YAML loads unchanged
through train_semi.py)."""
import copy


def cityscapes_semi(arch="resnet101", crop=769, batch_size=2, epochs=200, sync_bn=True, aux=True, num_classes=19):
    cfg = dict(
        dataset=dict(type="cityscapes_semi", batch_size=batch_size, n_sup=744, ignore_label=255,
                     train=dict(crop=dict(type="rand", size=[crop, crop]))),
        trainer=dict(
            epochs=epochs, eval_on=True, sup_only_epoch=0,
            optimizer=dict(type="SGD", kwargs=dict(lr=0.01, momentum=0.9, weight_decay=0.0005)),
            lr_scheduler=dict(mode="poly", kwargs=dict(power=0.9)),
            unsupervised=dict(TTA=False, drop_percent=80, apply_aug="cutmix"),
            contrastive=dict(negative_high_entropy=True, low_rank=3, high_rank=20, current_class_threshold=0.3,
                             current_class_negative_threshold=1, unsupervised_entropy_ignore=80,
                             low_entropy_threshold=20, num_negatives=50, num_queries=256, temperature=0.5),
        ),
        criterion=dict(type="ohem", kwargs=dict(thresh=0.7, min_kept=100000)),
        net=dict(
            num_classes=num_classes, sync_bn=sync_bn, ema_decay=0.99,
            encoder=dict(type=f"u2pl.models.resnet.{arch}",
                         kwargs=dict(multi_grid=True, zero_init_residual=True, fpn=True,
                                     replace_stride_with_dilation=[False, True, True], pretrained=False)),
            decoder=dict(type="u2pl.models.decoder.dec_deeplabv3_plus",
                         kwargs=dict(inner_planes=256, dilations=[12, 24, 36])),
        ),
    )
    if aux:
        cfg["net"]["aux_loss"] = dict(aux_plane=1024, loss_weight=0.4)
    return copy.deepcopy(cfg)


def pascal_semi(arch="resnet101", crop=513, batch_size=4, epochs=80, sync_bn=True):
    """experiments/pascal/1464/ours/config.yaml (BASELINE configs[2] family): C=21, no aux head, plain CE,
    lr 0.001 with x10 on the heads (train_semi.py:100-110), sup_only_epoch default 1."""
    cfg = cityscapes_semi(arch=arch, crop=crop, batch_size=batch_size, epochs=epochs, sync_bn=sync_bn, aux=False,
                          num_classes=21)
    cfg["dataset"].update(type="pascal_semi", n_sup=1464)
    cfg["trainer"].pop("sup_only_epoch")          # reference default: 1
    cfg["trainer"]["optimizer"]["kwargs"].update(lr=0.001, weight_decay=0.0001)
    cfg["criterion"] = dict(type="CELoss", kwargs=dict(use_weight=False))
    return cfg
