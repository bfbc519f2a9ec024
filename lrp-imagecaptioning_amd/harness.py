"""Explanation harness with the call surface of explain_image.Explainer (explain_image.py:4-183),
returning arrays instead of writing matplotlib figures: per word the raw relevance
(224,224,3), the seismic heat-map, the attention map and `r_words`."""
import os

import numpy as np

from .postprocess import heatmap, postprocess

VGG_BGR_MEAN = np.array([103.939, 116.779, 123.68], dtype=np.float32)


class ImagePreprocessor(object):
    """models/preprocessors.py:10-53 for the vgg16 encoder: 224x224, RGB->BGR, mean subtraction
    (keras.applications.vgg16.preprocess_input, 'caffe' mode)."""
    IMAGE_SIZE = (224, 224)

    def __init__(self, encoder="vgg16"):
        if encoder not in ("vgg16", "vgg19"):
            raise NotImplementedError("do not have this encoder option")
        self.encoder = encoder

    def _preprocess_an_image(self, img_path):
        from PIL import Image
        img = Image.open(img_path).convert("RGB").resize(self.IMAGE_SIZE[::-1], Image.NEAREST)   # keras load_img default
        arr = np.asarray(img, dtype=np.float32)
        return arr[..., ::-1] - VGG_BGR_MEAN

    def preprocess_images(self, img_paths, random_transform=False):
        return [self._preprocess_an_image(p) for p in img_paths]

    def preprocess_batch(self, img_list):
        return np.array(img_list)

    def preprocess_on_device(self, rgb_u8):
        """The same transformation for a batch of decoded RGB images already on the GPU ((NB, H0, W0, 3) uint8 tensor):
        lrp_preprocess_images -> (NB, 224, 224, 3) float32 tensor, ready for lrp_encode_images."""
        from .engine import preprocess_images
        return preprocess_images(rgb_u8, self.IMAGE_SIZE)


class Explainer(object):
    def __init__(self, model, weight_path, explainer, max_caption_length, beam_size):
        self._img_encoder = model.img_encoder
        self._image_preprocessor = explainer._dataset_provider.image_preprocessor or ImagePreprocessor(self._img_encoder)
        self._caption_preprocessor = explainer._dataset_provider.caption_preprocessor
        self._explainer = explainer
        self._max_caption_length = max_caption_length
        self._beam_size = beam_size
        if self._img_encoder in ["vgg16", "vgg19"]:                      # explain_image.py:17-20
            self._color_conversion = "BGRtoRGB"
            self._reshape_size = (14, 14)
            self._upscale = 16
        else:
            raise NotImplementedError("the img_encode is not valid, [vgg16, vgg19, inception_v3]")

    def _project(self, x):
        """explain_image.py:27-34."""
        absmax = np.max(np.abs(x))
        x = 1.0 * x / absmax
        if np.sum(x < 0):
            x = (x + 1) / 2
        return x * 255

    def _preprocess_img(self, img_path):
        imgs = self._image_preprocessor.preprocess_images(img_path)
        return (self._caption_preprocessor.SOS_TOKEN_LABEL_ENCODED, self._image_preprocessor.preprocess_batch(imgs))

    def _predict_caption(self, X):
        return self._explainer._beam_search(X, beam_size=self._beam_size)[0]

    def _explain_captions(self, X, captions, save_folder=None):
        """explain_image.py:45-87 without the plotting."""
        self._explainer._forward_beam_search(X, captions)
        img_encode_relevance, attention = self._explainer._explain_sentence()
        _, img_input = X
        rel, hms = [], []
        if getattr(self._explainer, "_batched_cnn", False) and len(attention):
            # every word of the caption in one walk of the image model, heat-maps rendered on the device
            from .engine import heatmap_render
            stack = np.concatenate([np.asarray(r, dtype=np.float32) for r in img_encode_relevance[:len(attention)]], axis=0)
            dev = self._explainer._explain_CNN(img_input, stack, as_tensor=True)
            hms = heatmap_render(dev, color_conversion=self._color_conversion).cpu().numpy()
            rel = dev.cpu().numpy()
        else:
            for i in range(len(attention)):
                relevance = self._explainer._explain_CNN(img_input, img_encode_relevance[i])
                hp = postprocess(relevance, self._color_conversion, False)
                rel.append(relevance[0])
                hms.append(heatmap(hp)[0])
        res = {"captions": list(captions), "relevance": np.asarray(rel), "heatmaps": np.asarray(hms),
               "attention": np.asarray(attention), "r_words": self._explainer.r_words}
        if save_folder:
            os.makedirs(save_folder, exist_ok=True)
            np.savez_compressed(os.path.join(save_folder, "lrp_hm.npz"), **{k: v for k, v in res.items() if v is not None})
        return res

    def _explain_single_word(self, X, captions, save_folder, t):
        """explain_image.py:123-150."""
        self._explainer._forward_beam_search(X, captions)
        _, img_input = X
        R, attention = self._explainer._explain_lstm_single_word_sequence(t)
        relevance = self._explainer._explain_CNN(img_input, R)
        hp = heatmap(postprocess(relevance, self._color_conversion, False))[0]
        return {"relevance": relevance[0], "heatmap": self._project(hp), "attention": attention,
                "r_words": self._explainer.r_words}

    def analyze_img(self, folder, img_path):
        self.img_path = img_path
        X = self._preprocess_img([img_path])
        captions = self._predict_caption(X)
        save_folder = os.path.join(folder, os.path.basename(img_path)) if folder else None
        return self._explain_captions(X, captions, save_folder)

    def analyze_single_word(self, folder, img_path, t):
        self.img_path = img_path
        X = self._preprocess_img([img_path])
        captions = self._predict_caption(X)
        return self._explain_single_word(X, captions, folder, t)
