(ns raytracing.gpu
  "librtmi.so bound with the Foreign Function & Memory API (JDK 22 or later; run the JVM with
  --enable-native-access=ALL-UNNAMED). render-gpu takes the place of the thread-per-row sample loop of
  core.clj:96-127 and returns the packed pixels that loop hands to setRGB (core.clj:123-126).

  The world goes to the library as one scene-file blob (raytracing.flatten/scene-bytes), so the only
  structs written here are rt_camera (24 doubles) and rt_params (64 bytes). Their offsets below are
  those of include/rtmi.h; tests/test_clojure_binding.py holds them against the C structs."
  (:require [raytracing.flatten :as flatten])
  (:import (java.lang.foreign Arena FunctionDescriptor Linker MemorySegment SymbolLookup ValueLayout)
           (java.lang.invoke MethodHandle)
           (java.nio ByteBuffer)))

;; ---- struct layouts of include/rtmi.h (little-endian, natural alignment, LP64)
(def camera-bytes 192)
(def camera-offsets
  "rt_camera: the camera map of camera.clj:24-33, field for field."
  {:origin 0 :lower-left-corner 24 :horizontal 48 :vertical 72 :u 96 :v 120 :w 144
   :lens-radius 168 :time0 176 :time1 184})
(def params-bytes 64)
(def params-offsets
  {:width 0 :height 4 :spp 8 :sample-offset 12 :max-depth 16 :shard-index 20 :shard-count 24 :flags 28
   :background 32 :seed 56})
(def desc-bytes 208)
(def desc-offsets
  "rt_scene_desc: counts are int32, everything else a pointer."
  {:abi-version 0 :n-prims 4 :prim-type 8 :prim-d 16 :prim-mat 24
   :n-nodes 32 :node-kind 40 :node-i 48 :node-d 56 :n-children 64 :node-children 72 :root 80 :n-media 84
   :n-mats 88 :mat-type 96 :mat-d 104 :mat-tex 112
   :n-tex 120 :tex-type 128 :tex-d 136 :tex-i 144
   :n-perlin 152 :perlin-ranvec 160 :perlin-perm 168
   :n-images 176 :img-w 184 :img-h 192 :img-rgb8 200})
(def split-tiles 0)
(def split-samples 1)
(def default-seed 0x5EED0001)

;; ---- downcalls
(def ^:private ^Linker linker (Linker/nativeLinker))
(def ^:private ^SymbolLookup lib
  (SymbolLookup/libraryLookup (or (System/getenv "RTMI_LIB") "librtmi.so") (Arena/global)))
(def ^:private ADDR ValueLayout/ADDRESS)
(def ^:private I32 ValueLayout/JAVA_INT)
(def ^:private I64 ValueLayout/JAVA_LONG)
(def ^:private F64 ValueLayout/JAVA_DOUBLE)

(defn- down ^MethodHandle [^String sym ^FunctionDescriptor fd]
  (.downcallHandle linker
                   (.orElseThrow (.find lib sym))
                   fd
                   (make-array java.lang.foreign.Linker$Option 0)))

(defn- fd [ret & args]
  (FunctionDescriptor/of ret (into-array java.lang.foreign.MemoryLayout args)))
(defn- fd-void [& args]
  (FunctionDescriptor/ofVoid (into-array java.lang.foreign.MemoryLayout args)))

(def ^:private rt-init (down "rt_init" (fd I32 I32 ADDR)))
(def ^:private rt-shutdown (down "rt_shutdown" (fd-void ADDR)))
(def ^:private rt-last-error (down "rt_last_error" (fd ADDR)))
(def ^:private rt-scene-from-memory (down "rt_scene_create_from_memory" (fd I32 ADDR ADDR I64 ADDR)))
(def ^:private rt-scene-destroy (down "rt_scene_destroy" (fd-void ADDR)))
(def ^:private rt-render (down "rt_render" (fd I32 ADDR ADDR ADDR ADDR ADDR ADDR ADDR)))
(def ^:private rt-multi-init (down "rt_multi_init" (fd I32 ADDR I32 ADDR)))
(def ^:private rt-multi-shutdown (down "rt_multi_shutdown" (fd-void ADDR)))
(def ^:private rt-multi-last-error (down "rt_multi_last_error" (fd ADDR)))
(def ^:private rt-multi-scene-create (down "rt_multi_scene_create" (fd I32 ADDR ADDR ADDR)))
(def ^:private rt-multi-scene-destroy (down "rt_multi_scene_destroy" (fd-void ADDR)))
(def ^:private rt-multi-render (down "rt_multi_render" (fd I32 ADDR ADDR ADDR ADDR I32 ADDR ADDR ADDR)))
(def ^:private rt-image-decode-jpeg (down "rt_image_decode_jpeg" (fd I32 ADDR I64 ADDR ADDR ADDR I64)))
(def ^:private rt-image-decode (down "rt_image_decode" (fd I32 ADDR I64 ADDR ADDR ADDR I64)))

(defn- call [^MethodHandle h & args]
  (.invokeWithArguments h ^java.util.List (vec args)))

(defn- c-string [^MemorySegment p]
  (if (= 0 (.address p))
    ""
    (.getString (.reinterpret p 4096) 0)))

(defn check!
  "A negative return of an rt_* call -> ex-info carrying the library's own message."
  ([rc] (check! rc rt-last-error))
  ([rc ^MethodHandle last-error]
   (when (neg? (long rc))
     (throw (ex-info (c-string (call last-error)) {:code rc})))
   rc))

;; ---- the two structs
(defn- put-vec3! [^MemorySegment seg offset v]
  (doseq [[i x] (map-indexed vector v)]
    (.set seg ^java.lang.foreign.ValueLayout$OfDouble F64 (long (+ offset (* 8 i))) (double x))))

(defn write-camera!
  "The map `camera` returns (camera.clj:24-33) -> rt_camera."
  ^MemorySegment [^Arena arena cam]
  (let [seg (.allocate arena (long camera-bytes) 8)]
    (doseq [k [:origin :lower-left-corner :horizontal :vertical :u :v :w]]
      (put-vec3! seg (camera-offsets k) (cam k)))
    (doseq [k [:lens-radius :time0 :time1]]
      (.set seg ^java.lang.foreign.ValueLayout$OfDouble F64 (long (camera-offsets k)) (double (cam k))))
    seg))

(defn write-params!
  "rt_params for one call. shard-index / shard-count 0 / 1 = the whole frame."
  ^MemorySegment [^Arena arena {:keys [width height spp sample-offset max-depth shard-index shard-count flags background seed]
                                 :or {sample-offset 0 shard-index 0 shard-count 1 flags 0 seed default-seed}}]
  (let [seg (.allocate arena (long params-bytes) 8)
        put-int! (fn [k v] (.set seg ^java.lang.foreign.ValueLayout$OfInt I32 (long (params-offsets k)) (int v)))]
    (put-int! :width width)
    (put-int! :height height)
    (put-int! :spp spp)
    (put-int! :sample-offset sample-offset)
    (put-int! :max-depth max-depth)
    (put-int! :shard-index shard-index)
    (put-int! :shard-count shard-count)
    (put-int! :flags flags)
    (put-vec3! seg (params-offsets :background) background)
    (.set seg ^java.lang.foreign.ValueLayout$OfLong I64 (long (params-offsets :seed)) (long seed))
    seg))

(defn- scene-segment ^MemorySegment [world]
  (let [^ByteBuffer bb (flatten/scene-bytes (flatten/flatten-world world))]
    (MemorySegment/ofBuffer bb)))

(defn write-desc!
  "The flattener's arrays -> rt_scene_desc with its arrays beside it in the arena (the form
  rt_scene_create and rt_multi_scene_create take; one GPU can use the blob instead)."
  ^MemorySegment [^Arena arena desc]
  (let [seg (.allocate arena (long desc-bytes) 8)
        put-int! (fn [k v] (.set seg ^java.lang.foreign.ValueLayout$OfInt I32 (long (desc-offsets k)) (int v)))
        put-ptr! (fn [k ^MemorySegment p] (.set seg ^java.lang.foreign.AddressLayout ADDR (long (desc-offsets k)) p))
        ints! (fn [k ^ints a] (put-ptr! k (.allocateFrom arena ^java.lang.foreign.ValueLayout$OfInt I32 a)))
        doubles! (fn [k ^doubles a] (put-ptr! k (.allocateFrom arena ^java.lang.foreign.ValueLayout$OfDouble F64 a)))
        images (:images desc)
        img-ptrs (.allocate arena (long (* 8 (max 1 (count images)))) 8)]
    (put-int! :abi-version (:abi-version desc))
    (put-int! :n-prims (alength ^ints (:prim-type desc)))
    (ints! :prim-type (:prim-type desc))
    (doubles! :prim-d (:prim-d desc))
    (ints! :prim-mat (:prim-mat desc))
    (put-int! :n-nodes (alength ^ints (:node-kind desc)))
    (ints! :node-kind (:node-kind desc))
    (ints! :node-i (:node-i desc))
    (doubles! :node-d (:node-d desc))
    (put-int! :n-children (alength ^ints (:children desc)))
    (ints! :node-children (:children desc))
    (put-int! :root (:root desc))
    (put-int! :n-media (:n-media desc))
    (put-int! :n-mats (alength ^ints (:mat-type desc)))
    (ints! :mat-type (:mat-type desc))
    (doubles! :mat-d (:mat-d desc))
    (ints! :mat-tex (:mat-tex desc))
    (put-int! :n-tex (alength ^ints (:tex-type desc)))
    (ints! :tex-type (:tex-type desc))
    (doubles! :tex-d (:tex-d desc))
    (ints! :tex-i (:tex-i desc))
    (put-int! :n-perlin (quot (alength ^ints (:perm desc)) 768))
    (doubles! :perlin-ranvec (:ranvec desc))
    (ints! :perlin-perm (:perm desc))
    (put-int! :n-images (count images))
    (ints! :img-w (:img-w desc))
    (ints! :img-h (:img-h desc))
    (doseq [[i ^bytes im] (map-indexed vector images)]
      (.setAtIndex img-ptrs ^java.lang.foreign.AddressLayout ADDR (long i)
                   (.allocateFrom arena ValueLayout/JAVA_BYTE im)))
    (put-ptr! :img-rgb8 img-ptrs)
    seg))

;; ---- the render calls
(defn render-gpu
  "Drop-in for core.clj:96-127 on one GPU: -> int[] of width * height packed 0xFFRRGGBB, row 0 = the
  bottom row (y = 0 of core.clj:109), ready for setRGB. The call blocks for the whole render."
  [world cam {:keys [width height device] :or {device 0} :as opts}]
  (with-open [arena (Arena/ofConfined)]
    (let [blob (scene-segment world)
          cam-seg (write-camera! arena cam)
          prm (write-params! arena opts)
          pctx (.allocate arena ADDR)
          pscn (.allocate arena ADDR)
          argb (.allocate arena (long (* 4 width height)) 4)]
      (check! (call rt-init (int device) pctx))
      (let [ctx (.get pctx ^java.lang.foreign.AddressLayout ADDR 0)]
        (try
          (check! (call rt-scene-from-memory ctx blob (.byteSize blob) pscn))
          (let [scn (.get pscn ^java.lang.foreign.AddressLayout ADDR 0)]
            (try
              (check! (call rt-render ctx scn cam-seg prm MemorySegment/NULL argb MemorySegment/NULL))
              (.toArray argb ^java.lang.foreign.ValueLayout$OfInt I32)
              (finally (call rt-scene-destroy scn))))
          (finally (call rt-shutdown ctx)))))))

(defn render-multi-gpu
  "The same over several GPUs of one node (rt_multi_*: one host thread per device, the per-GPU sums
  combined by one RCCL reduce on the first device): what the reference's fan-out of rows over threads
  and its latch become (core.clj:109-128). split = split-tiles or split-samples."
  [world cam {:keys [width height devices split] :or {devices [0] split split-tiles} :as opts}]
  (with-open [arena (Arena/ofConfined)]
    (let [cam-seg (write-camera! arena cam)
          prm (write-params! arena opts)
          ids (.allocateFrom arena ^java.lang.foreign.ValueLayout$OfInt I32 (int-array devices))
          pm (.allocate arena ADDR)
          pms (.allocate arena ADDR)
          argb (.allocate arena (long (* 4 width height)) 4)
          multi-check! (fn [rc] (check! rc rt-multi-last-error))]
      (multi-check! (call rt-multi-init ids (int (count devices)) pm))
      (let [m (.get pm ^java.lang.foreign.AddressLayout ADDR 0)]
        (try
          ;; (rt_multi_scene_create takes the description struct, not the blob)
          (let [desc (write-desc! arena (flatten/flatten-world world))]
            (multi-check! (call rt-multi-scene-create m desc pms)))
          (let [ms (.get pms ^java.lang.foreign.AddressLayout ADDR 0)]
            (try
              (multi-check! (call rt-multi-render m ms cam-seg prm (int split) MemorySegment/NULL argb MemorySegment/NULL))
              (.toArray argb ^java.lang.foreign.ValueLayout$OfInt I32)
              (finally (call rt-multi-scene-destroy ms))))
          (finally (call rt-multi-shutdown m)))))))

(defn- decode-with
  [^MethodHandle entry ^bytes file-bytes]
  (with-open [arena (Arena/ofConfined)]
    (let [src (.allocateFrom arena ValueLayout/JAVA_BYTE file-bytes)
          pw (.allocate arena I32)
          ph (.allocate arena I32)
          n (long (alength file-bytes))]
      (check! (call entry src n pw ph MemorySegment/NULL (long 0)))
      (let [w (.get pw ^java.lang.foreign.ValueLayout$OfInt I32 0)
            h (.get ph ^java.lang.foreign.ValueLayout$OfInt I32 0)
            out (.allocate arena (long (* 3 w h)) 1)]
        (check! (call entry src n pw ph out (long (* 3 w h))))
        {:width w :height h :rgb8 (.toArray out ValueLayout/JAVA_BYTE)}))))

(defn decode-jpeg
  "(image-texture url)'s decode step by the library (texture.clj:68-81): the bytes of a JPEG file ->
  {:width w :height h :rgb8 byte[]} (top row first, R G B)."
  [^bytes file-bytes]
  (decode-with rt-image-decode-jpeg file-bytes))

(defn decode-image
  "The same for a JPEG, PNG, GIF or BMP file, told apart by the first bytes as ImageIO/read tells them apart
  (texture.clj:72)."
  [^bytes file-bytes]
  (decode-with rt-image-decode file-bytes))
