(ns raytracing.core-gpu
  "-main with the sample loop on the GPU. `(-main :samples n)` means what it means in raytracing.core
  (core.clj:50-51): the same constants, the same world, the same camera, the same window; only the
  body between the camera and display-image (core.clj:96-128) is one call into librtmi.so.
  Optional keys widen what core.clj:52-59 hard-codes: :scene :width :height :devices :split :seed.

  A maintainer who would rather patch raytracing.core itself replaces core.clj:109-128 (the dotimes
  over rows and the latch) by the `render-into!` call below and adds the keys to -main's argument map."
  (:require [raytracing.camera :refer [camera]]
            [raytracing.core :refer [display-image]]
            [raytracing.scenes :as scenes]
            [raytracing.gpu :as gpu])
  (:import (java.awt.image BufferedImage)))

(def worlds
  "core.clj:66-88: every world with the camera settings -main gives it. Unlike core.clj the worlds are
  built on demand (a zero-argument fn each), not all eight for every run."
  {:random-scene       {:world scenes/random-scene :aperture 0.1}
   :two-spheres        {:world scenes/two-spheres}
   :two-perlin-spheres {:world scenes/two-perlin-spheres}
   :earth              {:world scenes/earth}
   :simple-light       {:world scenes/simple-light :background [0.0 0.0 0.0] :lookfrom [26 3 6] :lookat [0 2 0]}
   :cornell-box        {:world scenes/cornell-box :background [0.0 0.0 0.0] :lookfrom [278 278 -800] :lookat [278 278 0] :vfov 40.0}
   :cornell-smoke      {:world scenes/cornell-smoke :lookfrom [278 278 -800] :lookat [278 278 0] :vfov 40.0}
   :final-scene        {:world scenes/final-scene :background [0.0 0.0 0.0] :lookfrom [478 278 -600] :lookat [278 278 0] :vfov 40.0}})

(defn render-into!
  "What core.clj:109-128 does with threads and a latch: row y of the render goes to image row
  (height + 30) - y, column 0 (core.clj:123-126)."
  [^BufferedImage image ^ints argb width height]
  (doseq [y (range height)]
    (.setRGB image 0 (int (- (+ (long height) 30) (long y))) (int width) 1 argb (int (* (long y) (long width))) (int width))))

(defn -main [& {:keys [samples scene width height devices split seed]
                :or   {samples 30 scene :final-scene width 200 devices [0] split gpu/split-tiles seed gpu/default-seed}}]
  (let [aspect-ratio (if height (/ (double width) (double height)) 1.0)
        image-width (long width)
        image-height (long (or height (/ image-width aspect-ratio)))
        max-depth 50
        {:keys [world lookfrom lookat vfov aperture background]
         :or   {lookfrom   [13 2 3]
                lookat     [0 0 0]
                aperture   0.0
                vfov       20.0
                background [0.70 0.80 1.0]}} (worlds scene)
        cam (camera lookfrom lookat [0 1 0] vfov aspect-ratio aperture 10.0 0.0 1.0)
        image (BufferedImage. (int (+ image-width 20)) (int (+ image-height 75)) BufferedImage/TYPE_INT_RGB)
        opts {:width image-width :height image-height :spp samples :max-depth max-depth
              :background background :seed seed :devices devices :split split :device (first devices)}
        argb (if (> (count devices) 1)
               (gpu/render-multi-gpu (world) cam opts)
               (gpu/render-gpu (world) cam opts))]
    (render-into! image argb image-width image-height)
    (display-image image image-width image-height)))
