(ns raytracing.flatten
  "The world vector that -main binds (core.clj:66-88) -> the flat description librtmi.so takes
  (include/rtmi.h, struct rt_scene_desc), as Java arrays and as ONE little-endian blob (the scene
  file of include/rtmi.h: rt_scene_create_from_memory / rt_scene_create_from_file).

  A post-order walk over the reference's records. Identity-keyed maps intern hittables, materials,
  textures, Perlin tables and images, so an object the world holds twice (the two children of a
  span-1 bvhnode, base.clj:147-149; one material on many spheres) is described once.
  raytracing_amd/flatten.py is the same walk over the Python mirror of these records; it is the
  one the tests run (this image has no JVM)."
  (:require [raytracing.base]
            [raytracing.rectangles]
            [raytracing.box]
            [raytracing.rotate]
            [raytracing.translate]
            [raytracing.constant-medium]
            [raytracing.materials]
            [raytracing.texture]
            [raytracing.noise])
  (:import (raytracing.base Sphere MovingSphere HittableList BvhNode)
           (raytracing.rectangles XYRect XZRect YZRect)
           (raytracing.box Box)
           (raytracing.rotate RotateY)
           (raytracing.translate Translate)
           (raytracing.constant_medium ConstantMedium)
           (raytracing.materials Lambertian Metal Dielectric DiffuseLight Isotropic)
           (raytracing.texture SolidColor CheckerTexture NoiseTexture ImageTexture)
           (java.util ArrayList IdentityHashMap)
           (java.nio ByteBuffer ByteOrder)))

;; ---- the numbers of include/rtmi.h (tests/test_clojure_binding.py holds them against the header's)
(def abi-version 2)
(def prim-stride 12)
(def node-istride 4)
(def node-dstride 6)
(def mat-dstride 4)
(def tex-dstride 4)
(def tex-istride 2)
(def prim-kinds {:sphere 0 :moving-sphere 1 :xy-rect 2 :xz-rect 3 :yz-rect 4})
(def node-kinds {:prim 0 :list 1 :bvh 2 :translate 3 :rotate-y 4 :medium 5})
(def mat-kinds {:lambertian 0 :metal 1 :dielectric 2 :diffuse-light 3 :isotropic 4})
(def tex-kinds {:solid 0 :checker 1 :noise 2 :image 3})
(def scene-magic "RTMISCN1")
(def scene-header-bytes 64)

;; ---- state of one walk: growing lists and the identity maps
(defn new-state []
  {:prim-type (ArrayList.) :prim-d (ArrayList.) :prim-mat (ArrayList.)
   :node-kind (ArrayList.) :node-i (ArrayList.) :node-d (ArrayList.) :children (ArrayList.)
   :mat-type (ArrayList.) :mat-d (ArrayList.) :mat-tex (ArrayList.)
   :tex-type (ArrayList.) :tex-d (ArrayList.) :tex-i (ArrayList.)
   :ranvec (ArrayList.) :perm (ArrayList.)
   :img-w (ArrayList.) :img-h (ArrayList.) :images (ArrayList.)
   :n-media (int-array 1)
   :node-of (IdentityHashMap.) :mat-of (IdentityHashMap.) :tex-of (IdentityHashMap.)
   :perlin-of (IdentityHashMap.) :image-of (IdentityHashMap.)})

(defn- add-all! [^ArrayList lst xs]
  (doseq [x xs] (.add lst x)))

(defn- padded
  "xs followed by zeros up to n entries."
  [xs n zero]
  (let [v (vec xs)]
    (when (> (count v) n)
      (throw (ex-info "record longer than its stride" {:values v :stride n})))
    (into v (repeat (- n (count v)) zero))))

;; ---- Perlin tables and images
(defn perlin!
  "noise.clj:35-37,72-83: 256 unit vectors and the three permutations of one (perlin) call."
  [st p]
  (let [^IdentityHashMap seen (:perlin-of st)]
    (or (.get seen p)
        (let [id (quot (.size ^ArrayList (:ranvec st)) (* 256 3))]
          (doseq [v (:ranvec p)]
            (add-all! (:ranvec st) (map double v)))
          (doseq [perm [(:perm-x p) (:perm-y p) (:perm-z p)]]
            (add-all! (:perm st) (map int perm)))
          (.put seen p id)
          id))))

(defn image!
  "texture.clj:74-81: (:data t) is a vector of rows, top row first, each a vector of [R G B]
  (the reference reverses ImageIO's B G R bytes with rseq). -> image id; the texels as one byte[]."
  [st t]
  (let [^IdentityHashMap seen (:image-of st)
        data (:data t)]
    (or (.get seen data)
        (let [id (.size ^ArrayList (:images st))
              w (int (:width t))
              h (int (:height t))
              out (byte-array (* 3 w h))]
          (doseq [[j row] (map-indexed vector data)
                  [i px] (map-indexed vector row)
                  [c v] (map-indexed vector px)]
            (aset-byte out (+ (* 3 (+ (* j w) i)) c) (unchecked-byte (long v))))
          (.add ^ArrayList (:img-w st) w)
          (.add ^ArrayList (:img-h st) h)
          (.add ^ArrayList (:images st) out)
          (.put seen data id)
          id))))

;; ---- textures and materials
(defn tex!
  "texture.clj:15-66 -> texture id."
  [st t]
  (let [^IdentityHashMap seen (:tex-of st)]
    (or (.get seen t)
        (let [[kind d i]
              (condp instance? t
                SolidColor     [(:solid tex-kinds) (:color-value t) [0 0]]
                CheckerTexture (let [odd (tex! st (:odd t))
                                     even (tex! st (:even t))]
                                 [(:checker tex-kinds) [] [odd even]])
                ;; (:scale t) is a ^long field: 4.0 is already 4 and 0.1 already 0 (texture.clj:38-39)
                NoiseTexture   [(:noise tex-kinds) [] [(perlin! st (:perlin-noise t)) (long (:scale t))]]
                ImageTexture   [(:image tex-kinds) [] [(if (:data t) (image! st t) -1) 0]]
                (throw (ex-info "not a texture" {:value t})))
              id (.size ^ArrayList (:tex-type st))]
          (.add ^ArrayList (:tex-type st) (int kind))
          (add-all! (:tex-d st) (padded (map double d) tex-dstride 0.0))
          (add-all! (:tex-i st) (padded (map int i) tex-istride (int 0)))
          (.put seen t id)
          id))))

(defn mat!
  "materials.clj:14-96 -> material id. Metal keeps its fuzz as stored (scenes.clj:60,65-67 call
  ->Metal directly, without the clamp of the `metal` constructor)."
  [st m]
  (let [^IdentityHashMap seen (:mat-of st)]
    (or (.get seen m)
        (let [[kind d tex]
              (condp instance? m
                Lambertian   [(:lambertian mat-kinds) [] (tex! st (:albedo m))]
                Metal        [(:metal mat-kinds) (conj (vec (:albedo m)) (:fuzz m)) -1]
                Dielectric   [(:dielectric mat-kinds) [(:ir m)] -1]
                DiffuseLight [(:diffuse-light mat-kinds) [] (tex! st (:emit m))]
                Isotropic    [(:isotropic mat-kinds) [] (tex! st (:albedo m))]
                (throw (ex-info "not a material" {:value m})))
              id (.size ^ArrayList (:mat-type st))]
          (.add ^ArrayList (:mat-type st) (int kind))
          (add-all! (:mat-d st) (padded (map double d) mat-dstride 0.0))
          (.add ^ArrayList (:mat-tex st) (int tex))
          (.put seen m id)
          id))))

;; ---- hittables
(defn- new-node! [st obj kind ni nd]
  (let [id (.size ^ArrayList (:node-kind st))]
    (.add ^ArrayList (:node-kind st) (int kind))
    (add-all! (:node-i st) (padded (map int ni) node-istride (int 0)))
    (add-all! (:node-d st) (padded (map double nd) node-dstride 0.0))
    (when obj (.put ^IdentityHashMap (:node-of st) obj id))
    id))

(defn prim-node!
  "A leaf: its parameters in the order of include/rtmi.h's RT_PRIM_* comments, its material."
  [st obj kind params mat]
  (let [pidx (.size ^ArrayList (:prim-type st))]
    (.add ^ArrayList (:prim-type st) (int kind))
    (add-all! (:prim-d st) (padded (map double (flatten params)) prim-stride 0.0))
    (.add ^ArrayList (:prim-mat st) (int (mat! st mat)))
    (new-node! st obj (:prim node-kinds) [pidx] [])))

(defn list-node!
  "HittableList, the sides of a Box, the world vector: child ids in the reference's scan order."
  [st obj child-ids]
  (let [first-child (.size ^ArrayList (:children st))]
    (add-all! (:children st) (map int child-ids))
    (new-node! st obj (:list node-kinds) [first-child (count child-ids)] [])))

(defn bvh-node! [st obj left right box]
  (new-node! st obj (:bvh node-kinds) [left right] (concat (:minimum box) (:maximum box))))

(defn wrap-node!
  "Translate (node_d = offset xyz) and RotateY (node_d = sin, cos)."
  [st obj kind child nd]
  (new-node! st obj kind [child] nd))

(defn medium-node! [st obj boundary phase neg-inv-density]
  (let [^ints n (:n-media st)
        m (aget n 0)]
    (aset n 0 (inc m))
    (new-node! st obj (:medium node-kinds) [boundary phase m] [neg-inv-density])))

(defn node!
  "Hittable record -> node id (base.clj, rectangles.clj, box.clj, rotate.clj, translate.clj,
  constant_medium.clj). The k of a rect is a ^long field: already truncated (rectangles.clj:13,50,87)."
  [st obj]
  (let [^IdentityHashMap seen (:node-of st)]
    (or (.get seen obj)
        (condp instance? obj
          Sphere         (prim-node! st obj (:sphere prim-kinds) [(:center obj) (:radius obj)] (:mat-ptr obj))
          MovingSphere   (prim-node! st obj (:moving-sphere prim-kinds)
                                     [(:center0 obj) (:center1 obj) (:time0 obj) (:time1 obj) (:radius obj)]
                                     (:mat-ptr obj))
          XYRect         (prim-node! st obj (:xy-rect prim-kinds) [(:x0 obj) (:x1 obj) (:y0 obj) (:y1 obj) (:k obj)] (:mp obj))
          XZRect         (prim-node! st obj (:xz-rect prim-kinds) [(:x0 obj) (:x1 obj) (:z0 obj) (:z1 obj) (:k obj)] (:mp obj))
          YZRect         (prim-node! st obj (:yz-rect prim-kinds) [(:y0 obj) (:y1 obj) (:z0 obj) (:z1 obj) (:k obj)] (:mp obj))
          HittableList   (list-node! st obj (mapv #(node! st %) (:objects obj)))
          Box            (list-node! st obj (mapv #(node! st %) (:sides obj)))
          BvhNode        (let [left (node! st (:left obj))
                               right (node! st (:right obj))]
                           (bvh-node! st obj left right (:box obj)))
          Translate      (wrap-node! st obj (:translate node-kinds) (node! st (:ptr obj)) (:offset obj))
          RotateY        (wrap-node! st obj (:rotate-y node-kinds) (node! st (:ptr obj))
                                     [(:sin-theta obj) (:cos-theta obj)])
          ConstantMedium (let [boundary (node! st (:boundary obj))
                               phase (mat! st (:phase-function obj))]
                           (medium-node! st obj boundary phase (:neg-inv-density obj)))
          (throw (ex-info "not a hittable" {:value obj}))))))

;; ---- the finished description
(defn- ints-of ^ints [^ArrayList lst] (int-array (map int lst)))
(defn- doubles-of ^doubles [^ArrayList lst] (double-array (map double lst)))

(defn freeze
  "The lists as the arrays of rt_scene_desc, keyed by its field names."
  [st root]
  {:abi-version abi-version
   :prim-type (ints-of (:prim-type st)) :prim-d (doubles-of (:prim-d st)) :prim-mat (ints-of (:prim-mat st))
   :node-kind (ints-of (:node-kind st)) :node-i (ints-of (:node-i st)) :node-d (doubles-of (:node-d st))
   :children (ints-of (:children st)) :root (int root) :n-media (aget ^ints (:n-media st) 0)
   :mat-type (ints-of (:mat-type st)) :mat-d (doubles-of (:mat-d st)) :mat-tex (ints-of (:mat-tex st))
   :tex-type (ints-of (:tex-type st)) :tex-d (doubles-of (:tex-d st)) :tex-i (ints-of (:tex-i st))
   :ranvec (doubles-of (:ranvec st)) :perm (ints-of (:perm st))
   :img-w (ints-of (:img-w st)) :img-h (ints-of (:img-h st)) :images (vec (:images st))})

(defn flatten-world
  "world = the vector -main selects at core.clj:66-88 (ray-color wraps it in a HittableList for every
  ray, core.clj:23; here it is one LIST node, the root)."
  [world]
  (let [st (new-state)
        root (list-node! st nil (mapv #(node! st %) world))]
    (freeze st root)))

;; ---- the same as one blob (scene file)
(defn- pad8 [n] (bit-and (+ (long n) 7) (bit-not 7)))

(defn scene-bytes
  "The description as the scene file of include/rtmi.h: magic, abi_version, the nine counts, four
  reserved words, then from byte 64 the arrays in struct order and the images, each zero-padded to a
  multiple of 8 bytes. A direct little-endian buffer, ready for rt_scene_create_from_memory."
  ^ByteBuffer [desc]
  (let [arrays [(:prim-type desc) (:prim-d desc) (:prim-mat desc) (:node-kind desc) (:node-i desc)
                (:node-d desc) (:children desc) (:mat-type desc) (:mat-d desc) (:mat-tex desc)
                (:tex-type desc) (:tex-d desc) (:tex-i desc) (:ranvec desc) (:perm desc)
                (:img-w desc) (:img-h desc)]
        bytes-of (fn [a] (if (instance? (Class/forName "[D") a)
                           (* 8 (alength ^doubles a))
                           (* 4 (alength ^ints a))))
        size (+ scene-header-bytes
                (reduce + (map #(pad8 (bytes-of %)) arrays))
                (reduce + (map #(pad8 (alength ^bytes %)) (:images desc))))
        bb (.order (ByteBuffer/allocateDirect (int size)) ByteOrder/LITTLE_ENDIAN)
        align! (fn [] (.position bb (int (pad8 (.position bb)))))]
    (.put bb (.getBytes ^String scene-magic "US-ASCII"))
    (doseq [v [(:abi-version desc)
               (alength ^ints (:prim-type desc)) (alength ^ints (:node-kind desc))
               (alength ^ints (:children desc)) (:root desc) (:n-media desc)
               (alength ^ints (:mat-type desc)) (alength ^ints (:tex-type desc))
               (quot (alength ^ints (:perm desc)) 768) (count (:images desc))
               0 0 0 0]]
      (.putInt bb (int v)))
    (doseq [a arrays]
      (if (instance? (Class/forName "[D") a)
        (doseq [x ^doubles a] (.putDouble bb (double x)))
        (doseq [x ^ints a] (.putInt bb (int x))))
      (align!))
    (doseq [^bytes im (:images desc)]
      (.put bb im)
      (align!))
    (.flip bb)
    bb))
